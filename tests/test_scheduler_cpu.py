"""Host logic of the run.py-style scheduler without a GPU: the class behind the SemanticNetwork boundary is the oracle-backed stand-in
(oracle/oracle_network.py), so only control flow is exercised here; tests/test_gpu_scheduler.py runs the same scenarios on the HIP
path."""
import pytest

from oracle.oracle_network import OracleSemanticNetwork
from sched_cases import case_asr_atr_control_loop, case_edge_pipeline_equals_synchronous_loop, case_other_scheduler_modes, case_reference_sampling_default, case_upload_period_is_the_train_period


@pytest.mark.parametrize("sampling", ["reference", "per_second"])
def test_asr_atr_control_loop(tmp_path, sampling):
    case_asr_atr_control_loop(tmp_path, OracleSemanticNetwork, sampling=sampling)


@pytest.mark.parametrize("sampling", ["reference", "per_second"])
def test_upload_period_is_the_train_period_not_the_send_period(tmp_path, sampling):
    case_upload_period_is_the_train_period(tmp_path, OracleSemanticNetwork, sampling=sampling)


def test_default_sampling_is_the_reference_fraction(tmp_path):
    case_reference_sampling_default(tmp_path, OracleSemanticNetwork)


@pytest.mark.parametrize("mode", ["early", "pretrained", "horizon"])
def test_other_scheduler_modes(tmp_path, mode):
    case_other_scheduler_modes(tmp_path, mode, OracleSemanticNetwork)


def test_edge_pipeline_equals_synchronous_loop(tmp_path):
    case_edge_pipeline_equals_synchronous_loop(tmp_path, OracleSemanticNetwork)
