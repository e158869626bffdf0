"""Host-side decisions that need no GPU: the weight-streaming GEMM's decomposition chooser, the row-count
policy, the library-GEMM selection file, and that the product never falls back to the CPU."""
import csv

import pytest
import torch

from sglang_amd import kernels as K
from sglang_amd import tuning


@pytest.mark.parametrize("M", [1, 16, 64])
@pytest.mark.parametrize("N,Kd", [(28672, 4096), (4096, 14336), (6144, 4096), (4096, 4096), (128256, 4096),
                                  (512, 256), (9728, 896), (896, 4864 - 4864 % 128), (16, 128)])
def test_wstream_config_is_always_launchable(M, N, Kd):
    """Whatever the cost model prefers, the pair must be one the C entry point accepts."""
    for need_combine in (False, True):
        nw, s = K.choose_wstream_config(M, N, Kd, need_combine)
        assert 4 <= nw <= 8 and 1 <= s <= Kd // 128
    if N % 32 == 0:
        nw, s = K.choose_wstream_config(M, N, Kd, True, True)
        assert 2 <= nw <= 4 and s == 1


def test_wstream_config_fills_the_chip_on_the_bench_shapes():
    """The fitted model reproduces the measured optima (benchmarks/gemm_sweep.py, M = 64): whole 256-group rounds."""
    assert K.choose_wstream_config(64, 28672, 4096) == (7, 1)          # 1792 tiles = 256 x 7
    nw, s = K.choose_wstream_config(64, 4096, 14336)                    # 256 tiles: 64 groups x 4 splits
    assert (256 // nw) * s == 256
    nw, s = K.choose_wstream_config(64, 4096, 4096)
    assert (256 // nw) * s == 256
    assert K.choose_wstream_config(64, 28672, 4096, True, True) == (4, 1)   # one pass, gate + up tile per wave


@pytest.mark.parametrize("M", [1, 16, 64, 100])
@pytest.mark.parametrize("N,Kd", [(28672, 4096), (4096, 14336), (6144, 4096), (128256, 4096), (151936, 896), (112, 256), (16, 128)])
def test_wstream_decomposition_two_tile_waves_only_where_they_pay(M, N, Kd):
    """Two output tiles per wave: launches without split-K, at most 64 rows, N % 32 == 0 -- and a group size the
    two-tile kernels exist for; everything else keeps choose_wstream_config()'s pair."""
    nw, tpw, s = K.choose_wstream_decomposition(M, N, Kd)
    assert tpw in (1, 2)
    if tpw == 2:
        assert s == 1 and M <= 64 and N % 32 == 0 and 2 <= nw <= 4
    else:
        assert (nw, s) == K.choose_wstream_config(M, N, Kd)
    assert K.choose_wstream_decomposition(M, N, Kd, True)[2] == K.choose_wstream_config(M, N, Kd, True)[1]


def test_wstream_decomposition_on_the_bench_shapes():
    assert K.choose_wstream_decomposition(64, 128256, 4096) == (4, 2, 1)      # lm_head: 200 -> 174 us
    assert K.choose_wstream_decomposition(64, 4096, 14336)[1] == 1            # split-K shapes: one tile per wave
    assert K.choose_wstream_decomposition(64, 28672, 4096, True, True) == (4, 2, 1)
    # the one-tile interleaved silu form (256 x 7 waves for Llama-3's gate_up) measured equal to the two-tile one: selectable,
    # not the default (kernels.WSTREAM_SILU_INTERLEAVED)
    import unittest.mock as um

    with um.patch.object(K, "WSTREAM_SILU_INTERLEAVED", True):
        K.choose_wstream_decomposition.cache_clear()
        assert K.choose_wstream_decomposition(64, 28672, 4096, True, True) == (7, 1, 1)
        assert K.choose_wstream_decomposition(64, 57344, 8192, True, True) == (7, 1, 1)
        assert K.choose_wstream_decomposition(64, 14336, 4096, True, True)[1] == 2
    K.choose_wstream_decomposition.cache_clear()


def test_wstream_row_policy():
    assert K.wstream_supported(64, 4096, 4096) and K.wstream_supported(128, 4096, 4096)
    assert not K.wstream_supported(129, 4096, 4096) and not K.wstream_supported(0, 4096, 4096)
    assert not K.wstream_supported(8, 4100, 4096) and not K.wstream_supported(8, 4096, 4000)
    assert K.wstream_preferred(64, 128256, 4096)
    assert K.wstream_preferred(128, 4096, 14336) and not K.wstream_preferred(128, 28672, 4096)
    for M in (65, 128):                                                # beyond 64 rows: narrow groups only
        nw, s = K.choose_wstream_config(M, 4096, 4096)
        assert nw in (4, 5)
        assert K.choose_wstream_config(M, 14336, 4096, True, True)[0] == 2


def test_gemm_selection_file_is_wellformed_and_lookup_only_needs_a_gpu():
    rows = list(csv.reader(open(tuning.RESULTS)))
    validators = {r[1]: r[2] for r in rows if r[0] == "Validator"}
    assert validators["GCN_ARCH_NAME"].startswith("gfx950")
    entries = [r for r in rows if r[0].startswith("GemmTunableOp")]
    assert len(entries) >= 8 and all(r[0] == "GemmTunableOp_BFloat16_TN" and r[1].startswith("tn_") for r in entries)
    # the bench's warm-prefill gate_up shape at TP=1 is among them
    assert any(r[1].startswith("tn_28672_7680_4096_") for r in entries)
    if not torch.cuda.is_available():
        assert tuning.load_gemm_selections() is False                  # nothing to select on a CPU-only box


def test_kernels_refuse_cpu_tensors():
    x = torch.zeros((4, 256), dtype=torch.bfloat16)
    w = torch.zeros((64, 256), dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        K.wstream_gemm(x, w)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        K.rmsnorm(x, torch.ones(256, dtype=torch.bfloat16), 1e-5)


def test_moe_block_height_holds_one_and_a_half_average_experts():
    """choose_moe_block_m: an expert whose rows spill into a second row block streams its weights twice."""
    assert K.choose_moe_block_m(2, 8) == 16            # one token, top-2
    assert K.choose_moe_block_m(32, 8) == 16           # 4 rows per expert on average
    assert K.choose_moe_block_m(128, 8) == 32          # the 64-token Mixtral decode batch: 16 on average -> 24 -> 32
    assert K.choose_moe_block_m(256, 8) == 48
    assert K.choose_moe_block_m(8192, 8) == 64
    for pairs in range(1, 600, 7):
        for experts in (4, 8, 64, 256):
            assert K.choose_moe_block_m(pairs, experts) in (16, 32, 48, 64)


def test_req_token_arrays():
    """Req.origin_array / get_fill_ids (harness/engine.py): int64 arrays that track the lists they mirror."""
    from array import array

    from sglang_amd.harness.engine import Req
    from sglang_amd.mem_cache.radix_cache import RadixKey

    q = Req(0, [5, 6, 7, 8], 3)
    assert isinstance(q.origin_array, array) and q.origin_array.typecode == "q" and list(q.origin_array) == [5, 6, 7, 8]
    assert q.origin_array is q.origin_array                         # built once
    assert list(q.get_fill_ids()) == [5, 6, 7, 8]
    q.output_ids.append(9)
    assert list(q.get_fill_ids()) == [5, 6, 7, 8, 9] and q.seqlen == 5
    key = RadixKey(q.origin_array[: len(q.origin_input_ids) - 1], q.extra_key, q.cache_salt)
    assert len(key) == 3 and key.match(RadixKey([5, 6, 1])) == 2
    q.origin_input_ids = [1, 2]                                      # a replaced prompt is picked up
    assert list(q.origin_array) == [1, 2]


def test_bench_helpers():
    """bench.py's workload builder and byte / flop bookkeeping (no GPU needed for these)."""
    import importlib.util
    from pathlib import Path

    from sglang_amd.harness.models import CONFIGS

    spec = importlib.util.spec_from_file_location("bench_mod", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cfg = CONFIGS["llama-3-8b"]
    assert bench.p_lin(cfg) == 32 * (4096 * 6144 + 4096 * 4096 + 4096 * 28672 + 14336 * 4096)   # 6.98e9
    prompts = bench.build_prompts(cfg, 4, 16, 896, 128)
    assert len(prompts) == 4 and all(len(g) == 16 for g in prompts)
    assert all(len(p) == 1024 and p[:896] == g[0][:896] for g in prompts for p in g)             # shared system prompt
    assert prompts[0][0][:896] != prompts[1][0][:896] and prompts[0][0][896:] != prompts[0][1][896:]
    assert bench.build_prompts(cfg, 4, 16, 896, 128) == prompts                                   # seeded
    # decode weight bytes: SURVEY 8(d) W_act = (P_lin + hidden * vocab) * 2 B = 15.0 GB for the dense 8B model;
    # Mixtral counts the experts a 64-row batch is expected to hit (all 8 at 128 picks)
    assert bench.decode_weight_bytes(cfg, 64) == (bench.p_lin(cfg) + 4096 * 128256) * 2
    mix = CONFIGS["mixtral-8x7b"]
    per_expert = 3 * 4096 * 14336 * 2
    assert 7.9 * per_expert * 32 < bench.decode_weight_bytes(mix, 64) - (32 * (4096 * 6144 + 4096 * 4096 + 4096 * 8) + 4096 * 32000) * 2 <= 8 * per_expert * 32
    # committed PMC summary (profiles/r02_pmc.json) -> bytes per launch of the gate_up weight stream
    pmc = bench.load_pmc()
    t = bench.hbm_bytes(bench.pmc_kernel(pmc, "decode", "wstream_gemm_kernel<4, 4, 2"))
    assert t is None or 0.9 * 237e6 < t < 1.2 * 237e6
    assert bench.hbm_bytes({"FETCH_SIZE": 100.0, "WRITE_SIZE": 10.0}) == 210 * 1024 and bench.hbm_bytes({}) is None


def test_pool_format_is_read_off_the_pool_not_assumed():
    """The drop-in path hands HipAttnBackend the REFERENCE's MHATokenToKVPool (platform.get_mha_kv_pool_cls): its format
    comes from `dtype` / `page_size` / `use_hnd` (memory_pool.py:1636-1653,1816), fp8 scales from the layer's host floats
    (radix_attention.py:129-130) -- a k_scale device tensor is never converted (it would synchronise inside a capture)."""
    import types

    import torch

    from sglang_amd.layers.attention.hip_backend import pool_kernel_format

    ref_pool = types.SimpleNamespace(dtype=torch.float8_e4m3fn, store_dtype=torch.uint8, page_size=16, use_hnd=True)

    class _NoFloat(torch.Tensor):
        def __float__(self):
            raise AssertionError("float(k_scale tensor): a host sync")

    layer = types.SimpleNamespace(k_scale=torch.ones(1).as_subclass(_NoFloat), v_scale=torch.ones(1).as_subclass(_NoFloat),
                                  k_scale_float=0.5, v_scale_float=2.0)
    assert pool_kernel_format(ref_pool, layer) == dict(kv_fp8=True, k_scale=0.5, v_scale=2.0, page_size=16, hnd=True)
    unset = types.SimpleNamespace(k_scale=None, v_scale=None, k_scale_float=None, v_scale_float=None)
    assert pool_kernel_format(ref_pool, unset)["k_scale"] == 1.0
    bf16 = types.SimpleNamespace(dtype=torch.bfloat16, store_dtype=torch.bfloat16, page_size=1, use_hnd=True)
    assert pool_kernel_format(bf16, layer) == dict(kv_fp8=False, k_scale=1.0, v_scale=1.0, page_size=1, hnd=False)
    import pytest

    with pytest.raises(NotImplementedError):
        pool_kernel_format(types.SimpleNamespace(dtype=torch.float8_e5m2, page_size=1), None)


def test_decode_attention_routing_rule_for_window_and_cap_layers(monkeypatch):
    """The shared-prefix (cascade) decode kernel implements plain causal attention over any pool format; a layer with a
    sliding window or a logit soft cap (radix_attention.py:115-148) is routed to the paged decode kernel, which honours
    both -- per LAYER, inside one batch whose other layers keep the cascade path."""
    import types

    import torch

    from sglang_amd import kernels
    from sglang_amd.layers.attention import hip_backend as hb

    calls = []
    monkeypatch.setattr(kernels, "cascade_decode_attention", lambda *a, **k: calls.append(("cascade", sorted(k))))
    monkeypatch.setattr(kernels, "decode_attention", lambda *a, **k: calls.append(("paged", {x: k[x] for x in ("sliding_window", "logit_cap") if x in k})))
    pool = types.SimpleNamespace(dtype=torch.bfloat16, page_size=1, use_hnd=False, get_key_buffer=lambda i: torch.zeros((8, 2, 64), dtype=torch.bfloat16),
                                 get_value_buffer=lambda i: torch.zeros((8, 2, 64), dtype=torch.bfloat16))
    be = hb.HipAttnBackend.__new__(hb.HipAttnBackend)
    be.token_to_kv_pool, be.debug_flags = pool, 0
    be.req_to_token_pool = types.SimpleNamespace(req_to_token=torch.zeros((4, 16), dtype=torch.int32))
    be.forward_metadata = hb._Meta(torch.ones(3, dtype=torch.int32), cascade=object())
    fb = types.SimpleNamespace(req_pool_indices=torch.arange(3), out_cache_loc=torch.arange(3))

    def layer(**kw):
        return types.SimpleNamespace(tp_q_head_num=4, qk_head_dim=64, v_head_dim=64, layer_id=0, scaling=0.125, sliding_window_size=-1,
                                     logit_cap=0.0, k_scale_float=None, v_scale_float=None, **kw)

    q = torch.zeros((3, 4 * 64), dtype=torch.bfloat16)
    be.forward_decode(q, None, None, layer(), fb, save_kv_cache=False)
    win = layer(); win.sliding_window_size = 32
    be.forward_decode(q, None, None, win, fb, save_kv_cache=False)
    cap = layer(); cap.logit_cap = 30.0
    be.forward_decode(q, None, None, cap, fb, save_kv_cache=False)
    assert [c[0] for c in calls] == ["cascade", "paged", "paged"]
    assert calls[1][1] == {"sliding_window": 32} and calls[2][1] == {"logit_cap": 30.0}


def _named(class_name, **attrs):
    """An attribute bag whose class carries a reference class NAME (what fused_decode's form test reads)."""
    obj = type(class_name, (), {})()
    for k, v in attrs.items():
        setattr(obj, k, v)
    return obj


def test_all_reduce_switch_points_and_fused_layer_gates():
    """Pure host rules: (1) one-shot / two-stage switch of the xGMI all-reduce (custom_all_reduce.py:260-307: two ranks
    always one-shot, 512 KiB at four, 256 KiB at eight) -- a loopback communicator standing in for one rank of a TP job
    applies THAT job's switch; (2) the fused decode layer takes plain cos / sin-cache ropes only."""
    import types

    from sglang_amd import fused_decode
    from sglang_amd.distributed.xgmi_all_reduce import DEFAULT_MAX_BYTES, one_shot_limit

    assert one_shot_limit(2, DEFAULT_MAX_BYTES) == DEFAULT_MAX_BYTES
    assert one_shot_limit(4, DEFAULT_MAX_BYTES) == 512 * 1024 and one_shot_limit(8, DEFAULT_MAX_BYTES) == 256 * 1024
    assert one_shot_limit(1, DEFAULT_MAX_BYTES) == 256 * 1024          # (an unqualified loopback world is the most conservative)

    # layer_fusable rejects before it ever touches a kernel: missing pieces, quantised / biased-MLP projections, exotic ropes
    w = torch.zeros((8, 128), dtype=torch.bfloat16)
    lin = lambda **kw: types.SimpleNamespace(weight=w, quant_method=None, bias=None, **kw)    # noqa: E731
    rope = type("MRotaryEmbedding", (), {})()
    rope.is_neox_style, rope.rotary_dim = True, 64
    attn = _named("LlamaAttention", qkv_proj=lin(), o_proj=lin(), rotary_emb=rope, head_dim=64, num_heads=2, num_kv_heads=1, attn=None)
    mlp = _named("LlamaMLP", gate_up_proj=lin(), down_proj=lin())
    layer = _named("LlamaDecoderLayer", self_attn=attn, mlp=mlp)
    assert not fused_decode.layer_fusable(layer, 4)                     # CPU weights: _plain_linear says no first
    assert not fused_decode._plain_linear(types.SimpleNamespace(weight=None))
    assert not fused_decode.layer_fusable(types.SimpleNamespace(self_attn=None, mlp=None), 4)
    # the rope gate itself, with the projection gate out of the way
    import unittest.mock as um

    with um.patch.object(fused_decode, "_plain_linear", lambda l: True), \
            um.patch.object(fused_decode.kernels, "wstream_preferred", lambda *a: True):
        assert not fused_decode.layer_fusable(layer, 4)                 # MRotaryEmbedding: not this form
        good = type("RotaryEmbedding", (), {})()
        good.is_neox_style, good.rotary_dim = True, 64
        attn.rotary_emb = good
        assert not fused_decode.layer_fusable(layer, 4)                 # hidden = 128 is fine, but 8 gate_up rows % 32 != 0
        big = torch.zeros((64, 128), dtype=torch.bfloat16)
        mlp.gate_up_proj = types.SimpleNamespace(weight=big, quant_method=None, bias=None)
        assert fused_decode.layer_fusable(layer, 4)
        # the exact reference classes only -- a subclass that inherits the hooked model forward but changes the
        # attention math (Ministral3Attention scales q) is not this form ...
        sub = _named("Ministral3Attention", **vars(attn))
        assert not fused_decode.layer_fusable(_named("LlamaDecoderLayer", self_attn=sub, mlp=mlp), 4)
        assert not fused_decode.layer_fusable(_named("Ministral3DecoderLayer", self_attn=attn, mlp=mlp), 4)
        assert fused_decode.layer_fusable(_named("Qwen2DecoderLayer", self_attn=_named("Qwen2Attention", **vars(attn)),
                                                 mlp=_named("Qwen2MLP", **vars(mlp))), 4)
        good.is_neox_style = False
        assert not fused_decode.layer_fusable(layer, 4)                 # GPT-J style pairs: not this form
    # ... and a LoRA-wrapped projection (lora/layers.py:39-50: .weight aliases the base weight, no quant_method) is not a
    # plain linear: streaming the base weights would drop the adapter delta
    gpu_like = types.SimpleNamespace(dtype=torch.bfloat16, dim=lambda: 2, stride=lambda i: 1, is_cuda=True)
    assert fused_decode._plain_linear(types.SimpleNamespace(weight=gpu_like, quant_method=None))
    assert not fused_decode._plain_linear(types.SimpleNamespace(weight=gpu_like, base_layer=object(), set_lora=False))


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("rows,hidden,epilogue", [(64, 4096, 0), (256, 8192, 0), (512, 4096, 1), (256, 8192, 1),
                                                  (2048, 8192, 0), (100, 4096, 1), (8192, 4096, 0)])
def test_two_stage_all_reduce_work_is_spread_over_the_ranks(world, rows, hidden, epilogue):
    """at the automatic grid every workgroup runs ONE iteration for messages up to 8 MiB (<= 256 rows with
    the fused epilogue); the owner of a unit must therefore depend on the workgroup index, or rank 0 sums everything
    and every peer pulls the whole result from it (the one-shot kernel's link traffic plus a third barrier)."""
    import ctypes

    from sglang_amd import native

    lib = native.lib()
    units = (ctypes.c_int64 * world)()
    blocks = lib.sgl_amd_xgmi_two_stage_owner_units(rows, hidden, world, epilogue, 0, units)
    assert 1 <= blocks <= 256
    got = list(units)
    total = rows if epilogue else -(-(rows * hidden // 8) // 512)
    assert sum(got) == total
    # every rank sums its share to within one unit per workgroup iteration boundary (4 chunks per unrolled step)
    slack = 1 if epilogue else 4
    assert max(got) - min(got) <= slack * max(1, -(-total // (blocks * slack)) // world + 1), got
    assert max(got) <= -(-total // world) + slack * 2, got


def test_fused_decode_resolves_the_kv_pool_like_the_reference():
    """The reference's ForwardBatch carries no pools (they are reached through the active attention backend,
    forward_context.py:70-76); this package's harness batch carries `token_to_kv_pool`.  kv_pool_of() takes either."""
    import types

    from sglang_amd import fused_decode

    pool_a, pool_b = object(), object()
    assert fused_decode.kv_pool_of(types.SimpleNamespace(token_to_kv_pool=pool_a)) is pool_a
    assert fused_decode.kv_pool_of(types.SimpleNamespace(attn_backend=types.SimpleNamespace(token_to_kv_pool=pool_b))) is pool_b
    with pytest.raises(AttributeError):
        fused_decode.kv_pool_of(types.SimpleNamespace())


def test_fp8_kv_rows_are_accepted_as_bytes_or_as_the_references_fp8_view():
    """This package's pool stores e4m3 rows as uint8; the reference's pool hands out `k_buffer[...].view(torch.float8_e4m3fn)`
    (memory_pool.py:2288-2290).  Same bytes: the kernels' argument checks take both, and nothing else, for an fp8 pool."""
    import torch

    from sglang_amd import kernels

    raw = torch.zeros((4, 2, 64), dtype=torch.uint8)
    assert kernels._kv_rows_dtype_ok(raw, True) and kernels._kv_rows_dtype_ok(raw.view(torch.float8_e4m3fn), True)
    assert not kernels._kv_rows_dtype_ok(raw.view(torch.float8_e5m2), True) and not kernels._kv_rows_dtype_ok(torch.zeros(4, dtype=torch.bfloat16), True)
    assert kernels._kv_rows_dtype_ok(torch.zeros(4, dtype=torch.bfloat16), False) and not kernels._kv_rows_dtype_ok(raw, False)


def test_fused_decode_says_once_why_a_decode_forward_is_not_fused(caplog):
    """explain() names the first failing condition; the hook logs it once per reason and never raises."""
    import logging
    import types

    from sglang_amd import fused_decode

    mode = types.SimpleNamespace(is_decode=lambda: True)
    model = types.SimpleNamespace(layers=[types.SimpleNamespace(self_attn=None, mlp=None)], layers_to_capture=[], pp_group=None)
    fb = types.SimpleNamespace(forward_mode=mode, positions=None, batch_size=4)
    assert fused_decode.explain(model, fb) == "layer 0: no self_attn / mlp.gate_up_proj / mlp.down_proj"
    assert "not DECODE" in fused_decode.explain(model, types.SimpleNamespace(forward_mode=types.SimpleNamespace(is_decode=lambda: False)))
    fused_decode._SAID.clear()
    with caplog.at_level(logging.INFO, logger="sglang_amd"):
        fused_decode._say_once_why_not(model, fb, None)
        fused_decode._say_once_why_not(model, fb, None)
    assert sum("stays on the operator-by-operator path" in r.message for r in caplog.records) == 1
    fused_decode._SAID.clear()


def test_eager_decode_batches_beyond_the_captured_buffers_do_not_raise():
    """the reference's batches carry int64 seq_lens, so every decode goes through the backend's persistent int32
    buffer (max(max_bs, 256) entries) and the shared-prefix workspace sized at capture.  A decode batch larger than
    `--cuda-graph-max-bs` runs EAGERLY (decode_cuda_graph_runner.py:683-687 `cuda_graph_bs <= self.max_bs`): it must get its own
    conversion / workspace instead of killing the scheduler; only a CAPTURE beyond the buffers is an error."""
    import types

    from sglang_amd.layers.attention.hip_backend import HipAttnBackend

    be = HipAttnBackend.__new__(HipAttnBackend)
    be.device = torch.device("cpu")
    be._seq_i32, be._seq_src, be._seq_i32_in_graph = torch.zeros(256, dtype=torch.int32), None, True        # graphs captured at max_bs <= 256
    lens = torch.arange(5, 305, dtype=torch.int64)
    fb = types.SimpleNamespace(seq_lens=lens)
    got, src = be._seq_lens_i32(fb)                              # 300 running requests, eager
    assert src is None and got.dtype == torch.int32 and torch.equal(got.long(), lens) and be._seq_i32.numel() == 256
    with pytest.raises(RuntimeError, match="capturing a batch of 300"):
        be._seq_lens_i32(fb, in_capture=True)
    small = types.SimpleNamespace(seq_lens=lens[:64])
    buf, src = be._seq_lens_i32(small)                           # a replayed bucket: the persistent buffer, filled inside the graph
    assert src is small.seq_lens and buf.data_ptr() == be._seq_i32.data_ptr() and buf.numel() == 64
    i32 = types.SimpleNamespace(seq_lens=lens.int())
    assert be._seq_lens_i32(i32) == (i32.seq_lens, None)         # this package's own harness: int32 already
    # the shared-prefix workspace: graphs hold the one sized at capture, an eager batch beyond it gets another
    be.num_q_heads, be.head_dim, be.max_context_len = 8, 64, 256
    be._cascade_ws, be._cascade_ws_eager, be._cascade_in_graph = None, None, False
    ws8 = be._cascade_workspace(8, in_capture=True)
    be._cascade_in_graph = True
    assert be._cascade_workspace(4) is ws8 and be._cascade_workspace(8) is ws8
    eager = be._cascade_workspace(300)
    assert eager is not ws8 and eager.max_batch >= 300 and be._cascade_ws is ws8
    assert be._cascade_workspace(200) is eager                   # reused while it fits
    with pytest.raises(RuntimeError, match="capturing a batch of 300"):
        be._cascade_workspace(300, in_capture=True)


def test_cascade_policy_switches(monkeypatch):
    """The shared-prefix decode plan is chosen per server (it is baked into the captured decode graphs): on by default -- also
    without a radix cache, where it measured level with the plain kernel -- and overridable by the environment and by the runner's
    own attribute."""
    from types import SimpleNamespace as NS

    from sglang_amd.layers.attention.hip_backend import cascade_wanted

    monkeypatch.delenv("SGLANG_AMD_CASCADE", raising=False)
    assert cascade_wanted(NS()) is True                                                   # harness runner: no server_args
    assert cascade_wanted(NS(server_args=NS(disable_radix_cache=True))) is True
    assert cascade_wanted(NS(enable_cascade_attention=False)) is False
    monkeypatch.setenv("SGLANG_AMD_CASCADE", "0")
    assert cascade_wanted(NS(server_args=NS(disable_radix_cache=False))) is False
    assert cascade_wanted(NS(enable_cascade_attention=True)) is True                      # the attribute wins
    monkeypatch.setenv("SGLANG_AMD_CASCADE", "1")
    assert cascade_wanted(NS()) is True


def test_attention_backend_refuses_what_it_does_not_compute():
    """RadixAttention.forward hands the backend whatever the model passed (**kwargs, radix_attention.py:150-159): attention sinks
    (gpt_oss.py:496, granite.py:220), the MLA rope split, sparse index attention; layers carry Grok's xai_temperature_len and the
    cross-attention switch.  The gfx950 kernels compute none of these: the call is refused by name, before any launch."""
    import pytest
    from types import SimpleNamespace as NS

    from sglang_amd.layers.attention.hip_backend import HipAttnBackend

    layer = NS(qk_head_dim=128, v_head_dim=128, is_cross_attention=False, xai_temperature_len=-1)
    HipAttnBackend._refuse_unsupported(layer, {})
    HipAttnBackend._refuse_unsupported(layer, dict(sinks=None, k_rope=None))
    for name in ("sinks", "k_rope", "q_rope", "idx_q"):
        with pytest.raises(NotImplementedError, match=name):
            HipAttnBackend._refuse_unsupported(layer, {name: torch.zeros(4)})
    with pytest.raises(NotImplementedError, match="xai_temperature_len"):
        HipAttnBackend._refuse_unsupported(NS(qk_head_dim=128, v_head_dim=128, is_cross_attention=False, xai_temperature_len=1024), {})
    with pytest.raises(NotImplementedError, match="cross attention"):
        HipAttnBackend._refuse_unsupported(NS(qk_head_dim=128, v_head_dim=128, is_cross_attention=True), {})
    with pytest.raises(NotImplementedError, match="v_head_dim"):
        HipAttnBackend._refuse_unsupported(NS(qk_head_dim=192, v_head_dim=128, is_cross_attention=False), {})
    be = HipAttnBackend.__new__(HipAttnBackend)
    for fwd in (be.forward_extend, be.forward_decode):                   # the guard runs first: nothing else of the backend is touched
        with pytest.raises(NotImplementedError, match="sinks"):
            fwd(None, None, None, layer, None, sinks=torch.zeros(4))


def test_rows_the_elementwise_kernels_can_walk():
    """layers/activation.rows_vectorisable: the gate in front of the RMSNorm / SiluAndMul kernels (16-byte vectors, no hidden copy)."""
    from sglang_amd.layers.activation import rows_vectorisable

    x = torch.zeros(6, 64, dtype=torch.bfloat16)
    assert rows_vectorisable(x, 64) and rows_vectorisable(x[0], 64) and rows_vectorisable(x.view(2, 3, 64), 64)
    assert rows_vectorisable(x[:, :32], 32)                       # a column slice keeps whole-vector row strides
    assert not rows_vectorisable(x[:, 4:36], 32)                  # ... but not a 16-byte aligned base
    assert not rows_vectorisable(x[:, ::2], 32)                   # strided last dimension
    assert not rows_vectorisable(torch.zeros(6, 12, dtype=torch.bfloat16), 12)       # width outside the vector
    assert not rows_vectorisable(x.view(2, 3, 64)[:, :2], 64)     # 3-D view that is not contiguous: a reshape would copy
    assert not rows_vectorisable(x, 32) and not rows_vectorisable(torch.zeros((), dtype=torch.bfloat16), 1)


def test_fused_layer_row_gates_at_llama3_8b_shapes():
    """Which decode batches the fused layer owns at Llama-3-8B's projection shapes: up to 64 rows every projection streams; at
    65..128 rows the narrow ones still do and the wide gate_up is the library's (the hybrid form of fused_decode.decode_layer);
    beyond 128 rows nothing streams and the layer is the operator-by-operator one."""
    import types
    import unittest.mock as um

    from sglang_amd import fused_decode, kernels

    def lin(n, k):
        return types.SimpleNamespace(weight=torch.empty((n, k), dtype=torch.bfloat16, device="meta"), quant_method=None, bias=None)

    rope = type("RotaryEmbedding", (), {})()
    rope.is_neox_style, rope.rotary_dim = True, 128
    attn = _named("LlamaAttention", qkv_proj=lin(6144, 4096), o_proj=lin(4096, 4096), rotary_emb=rope, head_dim=128, num_heads=32, num_kv_heads=8, attn=None)
    layer = _named("LlamaDecoderLayer", self_attn=attn, mlp=_named("LlamaMLP", gate_up_proj=lin(28672, 4096), down_proj=lin(4096, 14336)))
    with um.patch.object(fused_decode, "_plain_linear", lambda l: True):          # (meta weights are not device weights)
        for rows in (1, 64, 65, 96, 128):
            assert fused_decode.layer_fusable(layer, rows), rows
        assert not fused_decode.layer_fusable(layer, 129)
        assert "qkv_proj" in fused_decode.layer_unfusable_reason(layer, 200)
    assert kernels.wstream_preferred(64, 28672, 4096) and not kernels.wstream_preferred(65, 28672, 4096)      # the hybrid's switch
    assert kernels.wstream_preferred(128, 6144, 4096) and kernels.wstream_preferred(128, 4096, 14336)
