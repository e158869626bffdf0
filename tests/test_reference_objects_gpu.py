"""The gfx950 backend and operator forwards driven with the REAL reference objects: no
look-alikes from sglang_amd/harness, no AST stand-in -- `ReqToTokenPool`, `MHATokenToKVPool`, `RadixAttention`,
`ForwardBatch.init_new` (forward_batch_info.py:705), `ForwardContext` / `get_attn_backend()` (forward_context.py:66),
the piecewise-prefill custom op `unified_attention_with_output` (radix_attention.py:405), `RMSNorm`, `RotaryEmbedding`
are the reference's own classes, imported from a staged copy of the reference sources (tests/golden/ref_objects.py; the
test skips when nothing is staged -- /root/reference does not exist on the GPU box).

What runs, in the order ModelRunner.forward runs it:
  ForwardBatch.init_new(batch, runner)   -> positions, extend_start_loc, extend_*_lens computed by the reference
  HipAttnBackend(runner)                 -> reads the reference runner's fields, the reference pools' buffers
  backend.init_forward_metadata(fb)
  RadixAttention.forward(q, k, v, fb)    -> the reference layer finds the backend through get_attn_backend(), the backend
                                            stores K / V through the REFERENCE pool's set_kv_buffer and attends
for a cold extend, a second extend over the cached prefix, decode steps (eager and inside a hipGraph through
init_forward_metadata_capture_cuda_graph / replay), the piecewise-prefill path, and TARGET_VERIFY; every output is
compared with the oracle's fp32 attention over the reference pool's own rows."""
import sys
import types
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
import ref_objects as R  # noqa: E402

BF = torch.bfloat16
Hq, Hkv, D, L = 8, 2, 128, 2


@pytest.fixture(scope="module")
def ref():
    root = R.ref_root()
    if root is None:
        pytest.skip("reference sources are not staged (python tests/golden/ref_objects.py --stage in the build container)")
    return R.install(root)


class _Batch:
    """What ForwardBatch.init_new reads of a ScheduleBatch: the fields a generation batch carries, None for the rest."""

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return None


def _batch(ref, mode, dev, *, input_ids, req_pool, seq_lens, out_loc, extend=None, prefix=None, spec_info=None):
    b = _Batch()
    b.forward_mode = mode
    b.seq_lens = torch.tensor(seq_lens, device=dev)
    b.seq_lens_cpu = torch.tensor(seq_lens)
    b.seq_lens_sum = int(sum(seq_lens))
    b.input_ids = input_ids
    b.req_pool_indices = torch.tensor(req_pool, device=dev)
    b.out_cache_loc = out_loc
    b.reqs, b.has_grammar, b.return_logprob = [], False, False
    b.spec_info = spec_info
    if extend is not None:
        b.extend_lens, b.prefix_lens, b.extend_num_tokens = list(extend), list(prefix), int(sum(extend))
        b.extend_logprob_start_lens = [0] * len(extend)
        b.is_extend_in_batch = True
    return b


def _runner(ref, dev, r2t, kv):
    mc = types.SimpleNamespace(model_is_mrope=False, get_num_attention_heads=lambda tp: Hq // tp, get_num_kv_heads=lambda tp: max(1, Hkv // tp),
                               num_attention_heads=Hq, num_key_value_heads=Hkv, head_dim=D)
    return types.SimpleNamespace(device=dev, is_draft_worker=False, lora_manager=None, prefill_attention_backend_str="torch_native",
                                 server_args=types.SimpleNamespace(), ngram_embedding_manager=types.SimpleNamespace(enabled=False),
                                 model_config=mc, ps=types.SimpleNamespace(attn_dcp_size=1, attn_dcp_rank=0),
                                 req_to_token_pool=r2t, token_to_kv_pool=kv, sliding_window_size=None, tp_size=1, attn_tp_size=1)


def _check(got, want, what):
    from oracle.layer_parity import ulp_stats

    st = ulp_stats(got.float().cpu(), want.float().cpu())
    # (a decode output here has 3 x 8 x 128 elements: 0.995 allows the handful of 2-ulp roundings of P that the 0.999 bar of
    # the large parity runs allows in proportion)
    assert st["frac_within_1ulp"] >= 0.995 and st["max_ulp"] <= 3.0, (what, st)


def test_backend_and_layers_on_the_references_own_objects(ref, device):
    from oracle import ops as oo
    from sglang_amd.layers.attention.hip_backend import HipAttnBackend

    mp, fbi, ra, fc = ref.memory_pool, ref.forward_batch_info, ref.radix_attention, ref.forward_context
    dev = "cuda"
    g = torch.Generator().manual_seed(4)
    with R.single_rank(ref):
        r2t = mp.ReqToTokenPool(8, 512, dev, False)
        kv = mp.MHATokenToKVPool(4096, 1, BF, Hkv, D, L, dev, False, enable_alt_stream=False)
        layers = [ra.RadixAttention(Hq, D, D ** -0.5, Hkv, i) for i in range(L)]
        runner = _runner(ref, dev, r2t, kv)
        backend = HipAttnBackend(runner)
        # the reference resolves the pools THROUGH the backend (forward_context.py:70-76)
        assert backend.token_to_kv_pool is kv and backend.req_to_token_pool is r2t
        B, pools = 3, [1, 4, 2]
        lens0 = [37, 130, 20]
        slots = (torch.randperm(4000, generator=g) + 1).to(torch.int32).to(dev)      # scattered slots, 0 is the padding slot
        cursor = 0
        ctx = fc.ForwardContext(attn_backend=backend)

        def rand(*shape):
            return (torch.randn(shape, generator=g) * 0.5).to(BF).to(dev)

        def attach(fb):                         # what ModelRunner does behind init_new (model_runner.py: the pools and the backend)
            fb.req_to_token_pool, fb.token_to_kv_pool, fb.attn_backend = r2t, kv, backend
            return fb

        def run_extend(extend, prefix, piecewise=False):
            nonlocal cursor
            T = sum(extend)
            loc = slots[cursor: cursor + T].to(torch.int64)
            cursor += T
            off = 0
            for b in range(B):
                r2t.req_to_token[pools[b], prefix[b]: prefix[b] + extend[b]] = loc[off: off + extend[b]].to(torch.int32)
                off += extend[b]
            seq = [p + e for p, e in zip(prefix, extend)]
            batch = _batch(ref, fbi.ForwardMode.EXTEND, dev, input_ids=torch.zeros(T, dtype=torch.int64, device=dev), req_pool=pools,
                           seq_lens=seq, out_loc=loc, extend=extend, prefix=prefix)
            fb = attach(fbi.ForwardBatch.init_new(batch, runner, capture_hidden_mode=fbi.CaptureHiddenMode.NULL,
                                                  return_hidden_states_before_norm=False))
            # the reference computed the positions and the start offsets
            want_pos = torch.cat([torch.arange(p, p + e) for p, e in zip(prefix, extend)])
            assert torch.equal(fb.positions.cpu(), want_pos) and fb.extend_prefix_lens_cpu == list(prefix)
            backend.init_forward_metadata(fb)
            outs = []
            with fc.forward_context(ctx):
                for layer in layers:
                    q, k, v = rand(T, Hq * D), rand(T, Hkv * D), rand(T, Hkv * D)
                    if piecewise:
                        cm = ref.context_manager.set_tc_piecewise_forward_context(fb, layers, None, [], [], num_tokens=T, raw_num_tokens=T)
                        with cm:
                            o = layer(q, k, v, fb)                       # -> unified_attention_with_output (radix_attention.py:405)
                    else:
                        o = layer(q, k, v, fb)
                    # the rows went through the REFERENCE pool's set_kv_buffer
                    assert torch.equal(kv.get_key_buffer(layer.layer_id)[loc], k.view(T, Hkv, D))
                    assert torch.equal(kv.get_value_buffer(layer.layer_id)[loc], v.view(T, Hkv, D))
                    want = oo.extend_attention(q.view(T, Hq, D), kv.get_key_buffer(layer.layer_id), kv.get_value_buffer(layer.layer_id),
                                               r2t.req_to_token, torch.tensor(pools, device=dev), torch.tensor(seq, device=dev),
                                               torch.tensor(prefix, device=dev), torch.tensor(extend, device=dev), D ** -0.5, True, torch.float32)
                    _check(o.view(T, Hq, D), want, ("extend", prefix, layer.layer_id, piecewise))
                    outs.append(o)
            return seq

        seq = run_extend(lens0, [0, 0, 0])                                # cold prefill
        seq = run_extend([16, 70, 33], seq)                               # a second chunk over the cached prefix
        seq = run_extend([8, 8, 40], seq, piecewise=True)                 # the piecewise-prefill custom-op path

        def run_decode(graph=None):
            nonlocal cursor, seq
            loc = slots[cursor: cursor + B].to(torch.int64)
            cursor += B
            for b in range(B):
                r2t.req_to_token[pools[b], seq[b]] = loc[b].to(torch.int32)
            seq = [s + 1 for s in seq]
            batch = _batch(ref, fbi.ForwardMode.DECODE, dev, input_ids=torch.zeros(B, dtype=torch.int64, device=dev), req_pool=pools,
                           seq_lens=seq, out_loc=loc)
            fb = attach(fbi.ForwardBatch.init_new(batch, runner, capture_hidden_mode=fbi.CaptureHiddenMode.NULL,
                                                  return_hidden_states_before_norm=False))
            assert torch.equal(fb.positions.cpu(), torch.tensor(seq) - 1)          # clamp_position(seq_lens), the reference's
            backend.init_forward_metadata(fb)
            with fc.forward_context(ctx):
                for layer in layers:
                    q, k, v = rand(B, Hq * D), rand(B, Hkv * D), rand(B, Hkv * D)
                    o = layer(q, k, v, fb)
                    assert torch.equal(kv.get_key_buffer(layer.layer_id)[loc], k.view(B, Hkv, D))
                    want = oo.decode_attention(q.view(B, Hq, D), kv.get_key_buffer(layer.layer_id), kv.get_value_buffer(layer.layer_id),
                                               r2t.req_to_token, torch.tensor(pools, device=dev), torch.tensor(seq, device=dev), D ** -0.5,
                                               torch.float32)
                    _check(o.view(B, Hq, D), want, ("decode", seq, layer.layer_id))
            return fb

        for _ in range(3):
            run_decode()

        # ---- ForwardMode.TARGET_VERIFY: a draft chain of 4 tokens per request scored in one forward; the reference's batch
        # carries the flat custom mask and draft_token_num in spec_info (triton_backend.py:860-919)
        nd = 4
        T = B * nd
        loc = slots[cursor: cursor + T].to(torch.int64)
        cursor += T
        for b in range(B):
            r2t.req_to_token[pools[b], seq[b]: seq[b] + nd] = loc[b * nd:(b + 1) * nd].to(torch.int32)
        blocks = []
        for b in range(B):
            m = torch.ones((nd, seq[b] + nd), dtype=torch.bool)
            m[:, seq[b]:] = torch.tril(torch.ones((nd, nd), dtype=torch.bool))
            blocks.append(m.flatten())
        mask = torch.cat(blocks).to(dev)
        spec = types.SimpleNamespace(custom_mask=mask, draft_token_num=nd,
                                     positions=torch.cat([torch.arange(seq[b], seq[b] + nd) for b in range(B)]).to(dev))
        batch = _batch(ref, fbi.ForwardMode.TARGET_VERIFY, dev, input_ids=torch.zeros(T, dtype=torch.int64, device=dev), req_pool=pools,
                       seq_lens=seq, out_loc=loc, spec_info=spec)
        fbv = attach(fbi.ForwardBatch.init_new(batch, runner, capture_hidden_mode=fbi.CaptureHiddenMode.NULL, return_hidden_states_before_norm=False))
        assert fbv.forward_mode.is_target_verify() and torch.equal(fbv.positions, spec.positions)
        backend.init_forward_metadata(fbv)
        with fc.forward_context(ctx):
            q, k, v = rand(T, Hq * D), rand(T, Hkv * D), rand(T, Hkv * D)
            o = layers[1](q, k, v, fbv)
        full = [s + nd for s in seq]
        mip = [0]
        for b in range(B):
            mip.append(mip[-1] + nd * full[b])
        want = oo.extend_attention(q.view(T, Hq, D), kv.get_key_buffer(1), kv.get_value_buffer(1), r2t.req_to_token,
                                   torch.tensor(pools, device=dev), torch.tensor(full, device=dev), torch.tensor(seq, device=dev),
                                   torch.tensor([nd] * B, device=dev), D ** -0.5, True, torch.float32, custom_mask=mask, mask_indptr=mip)
        _check(o.view(T, Hq, D), want, ("target_verify", seq))
        # (the draft rows are dropped again: the next steps continue from the verified context)

        # ---- the decode step inside a hipGraph, through the reference's capture / replay protocol (base_attn_backend.py:65-107:
        # init_cuda_graph_state, then init_forward_metadata_out_graph(fb, in_capture=True) before the capture,
        # init_forward_metadata_in_graph(fb) inside it, init_forward_metadata_out_graph(fb) before every replay)
        backend2 = HipAttnBackend(runner)
        backend2.init_cuda_graph_state(B, B)
        ctx2 = fc.ForwardContext(attn_backend=backend2)
        st_loc = torch.zeros(B, dtype=torch.int64, device=dev)
        st_q, st_k, st_v = rand(B, Hq * D), rand(B, Hkv * D), rand(B, Hkv * D)
        batch = _batch(ref, fbi.ForwardMode.DECODE, dev, input_ids=torch.zeros(B, dtype=torch.int64, device=dev), req_pool=pools,
                       seq_lens=seq, out_loc=st_loc)
        fbg = attach(fbi.ForwardBatch.init_new(batch, runner, capture_hidden_mode=fbi.CaptureHiddenMode.NULL, return_hidden_states_before_norm=False))
        fbg.attn_backend = backend2
        st_seq = fbg.seq_lens                                              # the graph's static buffers are the batch's own tensors
        backend2.init_forward_metadata_out_graph(fbg, in_capture=True)
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), fc.forward_context(ctx2):
            backend2.init_forward_metadata_in_graph(fbg)
            layers[0](st_q, st_k, st_v, fbg)                               # warm-up outside the capture (writes slot 0: the padding slot)
            side.synchronize()
            with torch.cuda.graph(graph, stream=side):
                backend2.init_forward_metadata_in_graph(fbg)
                out_g = layers[0](st_q, st_k, st_v, fbg)
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(2):
            loc = slots[cursor: cursor + B].to(torch.int64)
            cursor += B
            for b in range(B):
                r2t.req_to_token[pools[b], seq[b]] = loc[b].to(torch.int32)
            seq = [s + 1 for s in seq]
            st_seq.copy_(torch.tensor(seq)); st_loc.copy_(loc)
            fbg.seq_lens_cpu = torch.tensor(seq)
            st_q.copy_(rand(B, Hq * D)); st_k.copy_(rand(B, Hkv * D)); st_v.copy_(rand(B, Hkv * D))
            backend2.init_forward_metadata_out_graph(fbg)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(kv.get_key_buffer(0)[loc], st_k.view(B, Hkv, D))
            want = oo.decode_attention(st_q.view(B, Hq, D), kv.get_key_buffer(0), kv.get_value_buffer(0), r2t.req_to_token,
                                       torch.tensor(pools, device=dev), st_seq, D ** -0.5, torch.float32)
            _check(out_g.view(B, Hq, D), want, ("graph decode", seq))


def test_registered_operator_forwards_on_the_references_own_op_instances(ref, device):
    """The out-of-tree forwards plugin.load() registers (BaseFusedOp.register_oot_forward) are plain functions that take the
    REFERENCE's op instance as `self`: here they get real `RMSNorm` / `RotaryEmbedding` instances and must equal the
    instances' own forward_native."""
    from sglang_amd.layers import layernorm, rotary_embedding

    dev = "cuda"
    g = torch.Generator().manual_seed(9)
    with R.single_rank(ref):
        norm = ref.layernorm.RMSNorm(4096, eps=1e-5).to(dev)
        with torch.no_grad():
            norm.weight.copy_((1.0 + 0.1 * torch.randn(4096, generator=g)).to(norm.weight.dtype))
        norm = norm.to(BF)
        x = torch.randn((37, 4096), generator=g).to(BF).to(dev)
        res = torch.randn((37, 4096), generator=g).to(BF).to(dev)
        want = norm.forward_native(x.clone())
        got = layernorm.RMSNorm.forward(norm, x.clone())
        assert float((got.float() - want.float()).abs().max()) <= 2.0 ** -7 * float(want.float().abs().max())
        want_x, want_r = norm.forward_native(x.clone(), res.clone())
        got_x, got_r = layernorm.RMSNorm.forward(norm, x.clone(), res.clone())
        assert torch.equal(got_r, want_r)
        assert float((got_x.float() - want_x.float()).abs().max()) <= 2.0 ** -7 * float(want_x.float().abs().max())

        rope = ref.rotary_base.RotaryEmbedding(128, 128, 4096, 500000.0, True, BF).to(dev)
        pos = torch.randint(0, 4096, (29,), generator=g).to(dev)
        q = torch.randn((29, 8 * 128), generator=g).to(BF).to(dev)
        k = torch.randn((29, 2 * 128), generator=g).to(BF).to(dev)
        wq, wk = rope.forward_native(pos, q.clone(), k.clone())
        gq, gk = rotary_embedding.RotaryEmbedding.forward(rope, pos, q.clone(), k.clone())
        assert torch.equal(gq, wq) and torch.equal(gk, wk)
