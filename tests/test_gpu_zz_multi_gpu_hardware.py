"""GPU, more than one device: the multi-GPU paths on real hardware -- libsdfgpu_multi.so with one rank per GPU (RCCL send /
recv between devices) and sdf_tools_amd/slab.py under torch.distributed.run with backend nccl.  These tests light up by
themselves on a box with >= 2 GPUs and skip on a single-GPU one (where tests/test_gpu_multi.py and tests/test_gpu_slab.py
run the same code with every logical rank on device 0).  The file sorts behind the other GPU tests on purpose: no multi-GPU
box was available to the builder, so under `pytest -x` a first-contact failure here must not hide the rest of the suite."""
import numpy as np
import pytest

from sdf_tools_amd import capi, synth

pytestmark = pytest.mark.gpu


def _boxes(shape):
    nx, ny, nz = shape
    m = np.zeros(shape, np.uint8)
    m[nx // 10: nx // 10 + max(2, nx // 8), ny // 2: ny // 2 + max(2, ny // 6), : max(2, nz // 3)] = 1
    m[nx // 2: nx // 2 + max(2, nx // 5), ny // 8: ny // 8 + max(2, ny // 5), nz // 4: nz // 4 + max(2, nz // 4)] = 1
    return m


def _gpus():
    return capi.device_count()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_multi_rccl_one_rank_per_gpu(world, gpu):
    if _gpus() < world:
        pytest.skip("needs >= %d GPUs (this box has %d)" % (world, _gpus()))
    mg = capi.MultiSdfGpu(world, list(range(world)))
    try:
        shape = (16 * world + 16, 40, 64)
        cases = [
            ("dense", synth.bernoulli_mask(shape, 0.5, 1), False, dict(dense_certified=True, whole_lines=False)),
            ("mid", synth.bernoulli_mask(shape, 0.06, 2), False, dict(dense_certified=False)),
            ("far", _boxes(shape), False, dict(dense_certified=False, whole_lines=True)),
            ("far vb", _boxes(shape), True, dict(whole_lines=True)),
            ("empty", np.zeros(shape, np.uint8), False, dict(whole_lines=True)),
            ("odd shape", synth.bernoulli_mask((8 * world + 5, 18, 20), 0.02, 4), True, {}),
        ]
        for name, m, vb, expect in cases:
            got, ext = mg.build(m, 0.05, vb)
            path = mg.last_path()
            assert path["rccl"], (name, world, path)           # the messages really went through RCCL
            want, want_ext = gpu.build(m, 0.05, vb)            # the single-GPU ABI
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, world)
            assert ext == want_ext, (name, world, ext, want_ext)
            for k, v in expect.items():
                assert path[k] == v, (name, world, path)
    finally:
        mg.close()


def test_multi_rccl_1024x512x512_matches_single_gpu(gpu):
    """BASELINE configs[3]-sized slabs per rank on every GPU of the box: dense tier and far-field tier, == the single-GPU ABI."""
    world = min(8, _gpus())
    if world < 2:
        pytest.skip("needs >= 2 GPUs (this box has %d)" % _gpus())
    shape = (1024, 512, 512)
    mg = capi.MultiSdfGpu(world, list(range(world)))
    try:
        for name, m in (("dense", synth.bernoulli_mask(shape, 0.5, 1)), ("far", _boxes(shape))):
            got, ext = mg.build(m, 0.01)
            assert mg.last_path()["rccl"]
            want, want_ext = gpu.build(m, 0.01)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
            assert ext == want_ext, name
            del got, want
    finally:
        mg.close()


@pytest.mark.parametrize("world", [1, 2, 8])
def test_slab_builder_over_nccl_torchrun(world):
    """sdf_tools_amd/slab.py (what bench.py --gpus N runs), one process per GPU under torch.distributed.run, backend nccl."""
    import os
    import socket
    import subprocess
    import sys
    if _gpus() < world:
        pytest.skip("needs >= %d GPUs (this box has %d)" % (world, _gpus()))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "tools", "slab_nccl_check.py")]
    p = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and ("SLAB_NCCL_OK world=%d" % world) in p.stdout, p.stdout[-3000:]
