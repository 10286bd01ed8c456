// The per-channel arithmetic behind the second stages of the BN reductions (shared by k_elementwise.hip and k_xx_stats.hip).
#pragma once
#include "common.hpp"

namespace ams {

struct BnFwdFin {
    double n; const float* center; const float* gamma; const float* beta; float eps, one_minus_decay;
    float *moving_mean, *moving_var, *scale, *shift, *save_mean, *save_rstd;
    __device__ void operator()(int c, int /*C*/, double s0, double s1) const {
        const double ctr = center ? (double)center[c] : 0.0;
        const double d1 = s0 / n, d2 = s1 / n;
        double var = d2 - d1 * d1;
        if (var < 0) var = 0;
        const float mean = (float)(ctr + d1);
        const float varf = (float)var;
        const float rstd = 1.0f / sqrtf(varf + eps);
        const float sc = gamma[c] * rstd;
        scale[c] = sc;
        shift[c] = beta[c] - mean * sc;
        if (save_mean) { save_mean[c] = mean; save_rstd[c] = rstd; }
        if (moving_mean) {
            const float unbiased = (float)(var * (n / (n > 1.5 ? n - 1.0 : 1.0)));
            moving_mean[c] = moving_mean[c] - (moving_mean[c] - mean) * one_minus_decay;
            moving_var[c] = moving_var[c] - (moving_var[c] - unbiased) * one_minus_decay;
        }
    }
};
struct BnBwdFin {
    double n; const float* gamma; const float* mean; const float* rstd;
    float *coefA, *coefB, *coefC, *dgamma, *dbeta;
    __device__ void operator()(int c, int /*C*/, double sdy, double sdyx) const {
        const double A = (double)gamma[c] * rstd[c];
        const double k = A * (sdyx / n) * rstd[c];
        coefA[c] = (float)A;
        coefC[c] = (float)(-k);
        coefB[c] = (float)(-A * (sdy / n) + k * mean[c]);
        dgamma[c] = (float)sdyx;
        dbeta[c] = (float)sdy;
    }
};

}  // namespace ams
