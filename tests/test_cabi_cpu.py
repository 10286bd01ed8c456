"""The C-ABI library loads without a GPU and exports exactly what include/ams_hip.h declares (no compute calls here)."""
import re
import subprocess
from pathlib import Path

import pytest

from ams_amd import hip


def _header_functions(root):
    text = (root / "include" / "ams_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(ams_[a-z0-9_]+)\s*\(", text))
    names -= {"ams_allreduce_cb"}
    return names


def test_header_binding_and_library_agree(golden_dir):
    root = golden_dir.parent.parent
    declared = _header_functions(root)
    assert declared == set(hip.SIGNATURES), (declared ^ set(hip.SIGNATURES))
    lib = hip.lib()                                  # loads on a CPU-only host; raises if a symbol is missing
    out = subprocess.run(["nm", "-D", "--defined-only", str(hip.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (ams_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    assert exported - declared == set(), "library exports undeclared symbols: %s" % (exported - declared)
    assert lib.ams_abi_version() == hip.ABI_VERSION


def test_struct_layouts_match_header():
    import ctypes as C
    # ams_layer_desc: 7 x int32, float, 5 x int64 ; ams_student_config: see header
    assert C.sizeof(hip.LayerDesc) == 7 * 4 + 4 + 5 * 8
    assert C.sizeof(hip.StudentConfig) == (6 + 32 + 3) * 4 + 4 + 2 * 8 + 3 * 4 + 4      # incl. padding before/after the int64 pair
    assert hip.StudentConfig.n_trainable.offset % 8 == 0


def test_arena_size_query_runs_without_gpu():
    """ams_student_arena_bytes is pure host arithmetic: it must work here and scale with batch / trainable."""
    import ctypes as C
    from ams_amd import spec as S
    from ams_amd.engine import layer_table
    from ams_amd.spec import BN_DECAY, BN_EPS_FROZEN, PIXEL_SCALE
    lib = hip.lib()
    sp = S.build_spec()

    def need(batch, trainable, height=512):
        cfg = hip.StudentConfig()
        cfg.abi_version = hip.ABI_VERSION
        cfg.height, cfg.width, cfg.max_batch = height, 2 * height, batch
        cfg.num_classes, cfg.n_selected = 19, 6
        for i, c in enumerate([0, 1, 2, 10, 11, 13]):
            cfg.class_indices[i] = c
        cfg.n_layers, cfg.trainable, cfg.act_dtype = len(sp.layers), trainable, hip.DT_F32
        cfg.n_trainable, cfg.n_stats = sp.n_trainable, sp.n_stats
        cfg.bn_decay, cfg.bn_eps_frozen, cfg.pixel_scale = BN_DECAY, BN_EPS_FROZEN, PIXEL_SCALE
        n = C.c_size_t()
        hip.check(lib.ams_student_arena_bytes(C.byref(cfg), layer_table(sp), C.byref(n)))
        return n.value

    frozen1, frozen8, train8 = need(1, 0), need(8, 0), need(8, 1)
    assert 200e6 < frozen1 < 400e6            # 4 ping-pong buffers of the largest layer (50.6 MB) + weights
    assert frozen8 > 4 * frozen1
    assert 8e9 < train8 < 12e9                # z, a, da for all 55 layers at B=8, 512x1024 (~1.05 GB per frame)
    # bad configuration -> error code + message, no crash
    cfg = hip.StudentConfig()
    n = C.c_size_t()
    assert lib.ams_student_arena_bytes(C.byref(cfg), layer_table(sp), C.byref(n)) == -1
    assert b"ABI version" in lib.ams_last_error()


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ams_amd.engine import StudentEngine
    with pytest.raises(hip.AmsHipError):
        StudentEngine([0, 1], 32)


def test_product_never_touches_the_oracle(golden_dir):
    root = golden_dir.parent.parent
    for path in (root / "ams_amd").rglob("*.py"):
        src = path.read_text()
        assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), path


def test_kernel_attributes_are_tracked_per_device():
    """the one-time kernel attributes (dynamic LDS limit) are per DEVICE.  The bookkeeping behind func_allow_lds is
    pure host code, so a process with students on two GPUs can be played through here with mocked ordinals: the second device must
    be told to set the attribute too, a repeat on either device must not, a larger request must."""
    lib = hip.lib()
    key = 0x5EED0001
    assert lib.ams_debug_launch_table_needs_attr(0, key, 100 * 1024) == 1          # first launch on device 0 sets it
    assert lib.ams_debug_launch_table_needs_attr(0, key, 100 * 1024) == 0          # later launches there do not
    assert lib.ams_debug_launch_table_needs_attr(1, key, 100 * 1024) == 1          # a student on device 1: set again
    assert lib.ams_debug_launch_table_needs_attr(1, key, 90 * 1024) == 0           # a smaller request is covered
    assert lib.ams_debug_launch_table_needs_attr(0, key, 150 * 1024) == 1          # a larger one raises the limit
    assert lib.ams_debug_launch_table_needs_attr(0, key + 1, 100 * 1024) == 1      # per kernel


def test_no_launch_path_reads_the_environment():
    """tuning knobs are read once (runtime.hip); no getenv on a launch path, no function-local launch state."""
    import re
    src = Path(__file__).resolve().parent.parent / "ams_amd" / "csrc"
    for f in sorted(src.glob("k_*.hip")):
        text = f.read_text()
        assert "getenv" not in text, f.name
        assert not re.search(r"static\s+(bool|size_t|int)\s+(attr_set|attr_lds|slots|per_cu_cache)", text), f.name


def test_product_library_has_no_ablated_kernels():
    """kernels with loads / MFMAs / stores removed ("wrong results by design") and their AMS_*_ABL switches are compiled only
    under -DAMS_MEASURE into libams_hip_measure.so (make measure).  The product library carries the ABL = 0 instantiations alone, and an
    environment variable cannot select anything else."""
    hip.lib()
    out = subprocess.run(["nm", "-C", str(hip.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    kernels = set(re.findall(r"ams::(xdw_wreg_kernel<[^>]*>|first_block_walk_kernel<[^>]*>|pw_gemm_f16x3_l<[^>]*>)", out))
    assert kernels, "kernel symbols not found"
    for k in kernels:
        args = [a.strip() for a in k[k.index("<") + 1:-1].split(",")]
        if k.startswith("xdw_wreg_kernel"):
            abl = args[6] if len(args) > 6 else "0"
        elif k.startswith("first_block_walk_kernel"):
            abl = args[0]
        else:
            abl = args[6] if len(args) > 6 else "0"
        assert abl == "0", "ablated kernel in the product library: %s" % k
    src = (Path(__file__).resolve().parent.parent / "ams_amd" / "csrc" / "runtime.hip").read_text()
    product = re.sub(r"#ifdef AMS_MEASURE.*?(#else|#endif)", "", src, flags=re.S)       # what the product build compiles of the knob reader
    assert "AMS_XWR_ABL" in product and "ignored" in product                              # (it names them only to refuse them)
    assert "v.xwr_abl" not in product and "v.pwh_abl" not in product and "v.fb_abl" not in product
