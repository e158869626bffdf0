"""Build-time guard: the kernels whose residency is tuned to a register budget must not spill (hipcc's allocator is
one refactor away from scratch traffic in the middle of a load burst).  Cross-compiles for gfx950; no GPU needed."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _resource_usage(src: str, tmp_path):
    out = tmp_path / "k.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}", "--cuda-device-only", "-S",
           str(ROOT / "sglang_amd" / "csrc" / src), "-o", str(out)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    usage = {}
    name = None
    for line in out.read_text().splitlines():
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size):\s+(\d+)", line)
        if m:
            usage.setdefault("pending", {})[m.group(1)] = int(m.group(2))
        if name and "pending" in usage and line.strip().startswith(".wavefront_size"):
            usage[name] = usage.pop("pending")
            name = None
    usage.pop("pending", None)
    return usage


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_cascade_chunk_kernel_fits_five_workgroups_per_cu(tmp_path):
    usage = _resource_usage("cascade_attention.hip", tmp_path)
    chunk = {k: v for k, v in usage.items() if "cascade_chunk_kernel" in k}
    assert len(chunk) == 8, sorted(usage)                       # head dim {64, 128} x {bf16, fp8} x {NHD, HND}
    for name, u in chunk.items():
        # at the 96-register budget hipcc parks the row's partial-slot index (one value, written before the first
        # barrier, read back after the softmax) in scratch in some instances: tolerated; anything more lands inside the
        # load burst (a reload waits for every row in flight) and must fail here
        assert u["vgpr_spill_count"] <= 3 and u["sgpr_spill_count"] == 0, (name, u)
        if "ILi128ELb0ELb0E" in name:                           # the bf16 token-major D = 128 instance (the benchmarked one): none at all
            assert u["vgpr_spill_count"] == 0, (name, u)
        assert u["vgpr_count"] <= 96, (name, u)                 # 512 / 5 waves per SIMD, 8-register granules
        # five workgroups per CU: LDS is handed out in 1280-byte granules -- the 16 KiB image + the token-split units' side buffer
        # (8 KiB at D = 128) + 256 B of (max, sum) = 20 granules, five of them 128 000 of the 163 840 bytes
        assert u["group_segment_fixed_size"] <= 25600, (name, u)
    # the looping form (large request tables: a bounded grid walks the item list) runs four workgroups per CU and must not spill at all
    # (a five-per-CU instance parks ~29 registers in scratch and measured 34 against 26.7 us per layer: profiles/r05_exp1b_cascade_forms.json)
    loop = {k: v for k, v in usage.items() if "cascade_chunk_loop_kernel" in k}
    assert len(loop) == 8, sorted(usage)
    for name, u in loop.items():
        assert u["vgpr_spill_count"] == 0 and u["sgpr_spill_count"] == 0, (name, u)
        assert u["vgpr_count"] <= 128 and u["group_segment_fixed_size"] <= 25600, (name, u)


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_extend_eight_wave_kernels_keep_their_registers_and_lds(tmp_path):
    """The 8-wave kernels run two waves per SIMD (256 registers each) in one workgroup per CU.  The 32x32 kernel holds
    O^T (64), two score sets (64), Q^T (32), P (16), the staging rows (16) and three K fragments at ~245: a spill would
    sit between the matrix instructions; LDS = two K and two V images."""
    usage = _resource_usage("extend_attention.hip", tmp_path)
    for kernel, lds, n in (("extend_attention_pipe_kernel", 72 * 1024, 4), ("extend_attention_dbuf_kernel", 80 * 1024, 2)):
        inst = {k: v for k, v in usage.items() if kernel in k}
        assert len(inst) == n, sorted(usage)                    # head dim {64, 128} (x {bf16, e4m3 rows} for the 32x32 kernel)
        for name, u in inst.items():
            assert u["vgpr_spill_count"] == 0 and u["sgpr_spill_count"] == 0, (name, u)
            assert u["vgpr_count"] <= 256, (name, u)
            assert u["group_segment_fixed_size"] <= lds, (name, u)


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_general_extend_kernels_do_not_spill(tmp_path):
    """Every instance of the general extend kernel (fp8 / paged pools, sliding window, soft cap, custom mask; the 4-wave
    shapes of short extends): a spill would sit inside the KV-tile loop."""
    usage = _resource_usage("extend_attention.hip", tmp_path)
    gen = {k: v for k, v in usage.items() if "extend_attention_kernel" in k}
    assert len(gen) == 12, sorted(usage)                        # head dim {64, 128} x {41, 42, 82} x {bf16, fp8}
    for name, u in gen.items():
        assert u["vgpr_spill_count"] == 0 and u["sgpr_spill_count"] == 0, (name, u)


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_moe_256_tile_kernel_keeps_its_registers_and_lds(tmp_path):
    """moe_gemm256_kernel runs two waves per SIMD in one workgroup per CU: 128 accumulator registers + two fragment sets
    (64) + the eight LDS-DMA row pointers at ~213; a spill would sit between the products.  LDS = two 64 KiB stages."""
    usage = _resource_usage("moe_tiled_gemm.hip", tmp_path)
    k256 = {k: v for k, v in usage.items() if "moe_gemm256_kernel" in k}
    assert len(k256) == 1, sorted(usage)
    for name, u in k256.items():
        assert u["vgpr_spill_count"] == 0 and u["sgpr_spill_count"] == 0, (name, u)
        assert u["vgpr_count"] <= 256, (name, u)
        assert u["group_segment_fixed_size"] == 128 * 1024, (name, u)
    k128 = {k: v for k, v in usage.items() if "moe_tiled_gemm_kernel" in k}
    assert len(k128) == 1
    for name, u in k128.items():
        assert u["vgpr_spill_count"] == 0 and u["sgpr_spill_count"] == 0 and u["vgpr_count"] <= 128, (name, u)   # two workgroups per CU
