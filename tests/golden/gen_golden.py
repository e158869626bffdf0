"""Generate golden fixtures by running the REAL reference code (build container only).

    python tests/golden/gen_golden.py          # needs /root/reference

`import sglang` fails here (orjson, msgspec, zmq, torchvision ... are absent),
so this script installs an import hook that (a) stubs the missing third-party
packages, (b) skips every sglang package __init__, and (c) lets a reference
module that cannot be imported degrade to a permissive stub -- while the modules
we actually call (`REAL`) must import for real.  The arithmetic we record is
therefore the reference's own code running on torch CPU.

Outputs (committed): tests/golden/*.pt, tests/golden/*.json.
Nothing under tests/ reads /root/reference at test time.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import json
import random
import sys
import types
from array import array
from pathlib import Path
from unittest import mock

import torch

REF = Path("/root/reference/python")
OUT = Path(__file__).resolve().parent


# ----------------------------------------------------------------------------- import hook
class _AnyMeta(type):
    def __getattr__(cls, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Any


class _Any(metaclass=_AnyMeta):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and not k and callable(a[0]):
            return a[0]  # decorator pass-through
        return _Any()

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Any()

    def __class_getitem__(cls, i):
        return cls

    def __iter__(self):
        return iter(())

    def __bool__(self):
        return False

    def __or__(self, o):
        return self

    def __ror__(self, o):
        return self


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if (name.startswith("__") and name.endswith("__")) or name in ABSENT_ATTRS:
            raise AttributeError(name)
        # `from package import submodule` on a package whose __init__ was skipped: the attribute is the submodule, if there is one
        if SUBMODULE_ATTRS and self.__dict__.get("__path__"):
            full = f"{self.__name__}.{name}"
            if full in sys.modules:
                return sys.modules[full]
            if importlib.machinery.PathFinder.find_spec(full, self.__dict__["__path__"]) is not None:
                return importlib.import_module(full)
        return _Any


# ---- a working stand-in for msgspec.Struct (ref_model.py; MSGSPEC_EMULATION) ------------------------------------------------
# The reference declares its request / parameter / record types as msgspec Structs.  With msgspec merely stubbed their generated
# constructors are gone (fields keep their class-level defaults, keyword arguments vanish).  This is the documented behaviour of
# msgspec.Struct restated as far as the reference's classes use it IN PROCESS (fields from annotations in definition order,
# positional / keyword construction, defaults and default factories with fresh copies of empty mutable defaults, __post_init__,
# field-wise equality and repr, structs.replace / asdict / fields / force_setattr); the serialisers stay stubs.
MSGSPEC_EMULATION = False


def _make_msgspec():
    import abc
    import copy

    class _NoDefault:
        def __repr__(self):
            return "NODEFAULT"

    NODEFAULT = _NoDefault()

    class _Field:
        def __init__(self, default=NODEFAULT, default_factory=NODEFAULT, name=None):
            self.default, self.default_factory, self.name = default, default_factory, name

    def field(*, default=NODEFAULT, default_factory=NODEFAULT, name=None):
        return _Field(default, default_factory, name)

    class StructMeta(abc.ABCMeta):
        def __new__(mcls, name, bases, ns, **kwargs):
            cls = super().__new__(mcls, name, bases, ns)
            fields = {}
            for b in reversed(cls.__mro__[1:]):
                fields.update(b.__dict__.get("__struct_defaults_map__", {}))
            for fname, typ in ns.get("__annotations__", {}).items():
                if "ClassVar" in (typ if isinstance(typ, str) else getattr(typ, "__name__", repr(typ))):
                    continue
                default = ns.get(fname, NODEFAULT)
                fields[fname] = default
                if isinstance(default, _Field):
                    delattr(cls, fname)
            cls.__struct_defaults_map__ = fields
            cls.__struct_fields__ = tuple(fields)
            inherited = getattr(cls, "__struct_config__", None)
            cfg = dict(vars(inherited)) if inherited is not None else {}
            cfg.update(kwargs)
            cls.__struct_config__ = types.SimpleNamespace(**cfg)
            return cls

        def __init__(cls, name, bases, ns, **kwargs):
            super().__init__(name, bases, ns)

    def _default_of(d):
        if isinstance(d, _Field):
            if d.default_factory is not NODEFAULT:
                return d.default_factory()
            d = d.default
        if d is NODEFAULT:
            raise KeyError
        return copy.copy(d) if isinstance(d, (list, dict, set, bytearray)) else d

    class Struct(metaclass=StructMeta):
        def __init__(self, *args, **kwargs):
            names = type(self).__struct_fields__
            if len(args) > len(names):
                raise TypeError(f"{type(self).__name__}: too many positional arguments")
            vals = dict(zip(names, args))
            for k, v in kwargs.items():
                if k not in type(self).__struct_defaults_map__ or k in vals:
                    raise TypeError(f"{type(self).__name__}: unexpected or repeated argument {k!r}")
                vals[k] = v
            for n in names:
                if n not in vals:
                    try:
                        vals[n] = _default_of(type(self).__struct_defaults_map__[n])
                    except KeyError:
                        raise TypeError(f"{type(self).__name__}: missing required argument {n!r}") from None
                object.__setattr__(self, n, vals[n])
            post = getattr(self, "__post_init__", None)
            if post is not None:
                post()

        def __repr__(self):
            return f"{type(self).__name__}({', '.join(f'{n}={getattr(self, n)!r}' for n in type(self).__struct_fields__)})"

        def __eq__(self, other):
            if type(other) is not type(self):
                return NotImplemented
            return all(getattr(self, n) == getattr(other, n) for n in type(self).__struct_fields__)

        def __hash__(self):
            if getattr(type(self).__struct_config__, "frozen", False):
                return hash(tuple(getattr(self, n) for n in type(self).__struct_fields__))
            return id(self)

        def __copy__(self):
            new = object.__new__(type(self))
            for n in type(self).__struct_fields__:
                object.__setattr__(new, n, getattr(self, n))
            return new

        def __iter__(self):                         # array_like structs unpack like tuples nowhere in the path; keep explicit
            raise TypeError(f"{type(self).__name__} is not iterable")

    def replace(obj, **changes):
        new = obj.__copy__()
        for k, v in changes.items():
            if k not in type(obj).__struct_defaults_map__:
                raise TypeError(k)
            object.__setattr__(new, k, v)
        return new

    def asdict(obj):
        return {n: getattr(obj, n) for n in type(obj).__struct_fields__}

    def fields(obj):
        cls = obj if isinstance(obj, type) else type(obj)
        out = []
        for n, d in cls.__struct_defaults_map__.items():
            f = d if isinstance(d, _Field) else _Field(default=d)
            out.append(types.SimpleNamespace(name=n, encode_name=f.name or n, default=f.default, default_factory=f.default_factory,
                                             type=cls.__annotations__.get(n) if hasattr(cls, "__annotations__") else None,
                                             required=f.default is NODEFAULT and f.default_factory is NODEFAULT))
        return tuple(out)

    def force_setattr(obj, name, value):
        object.__setattr__(obj, name, value)

    m = _Stub("msgspec")
    m.__path__ = []
    st = _Stub("msgspec.structs")
    st.replace, st.asdict, st.fields, st.force_setattr = replace, asdict, fields, force_setattr
    st.astuple = lambda obj: tuple(getattr(obj, n) for n in type(obj).__struct_fields__)
    m.Struct, m.StructMeta, m.field, m.NODEFAULT, m.structs = Struct, StructMeta, field, NODEFAULT, st
    m.UNSET = type("UnsetType", (), {"__repr__": lambda self: "UNSET", "__bool__": lambda self: False})()
    m.UnsetType = type(m.UNSET)
    m.MsgspecError = type("MsgspecError", (Exception,), {})
    m.DecodeError = type("DecodeError", (m.MsgspecError,), {})
    m.ValidationError = type("ValidationError", (m.DecodeError,), {})
    m.EncodeError = type("EncodeError", (m.MsgspecError,), {})
    return m, st


class _MsgspecLoader(importlib.abc.Loader):
    def create_module(self, spec):
        if "msgspec" not in _MSGSPEC:
            _MSGSPEC["msgspec"], _MSGSPEC["msgspec.structs"] = _make_msgspec()
        return _MSGSPEC.get(spec.name) or _Stub(spec.name)

    def exec_module(self, m):
        if not hasattr(m, "__path__"):
            m.__path__ = []


_MSGSPEC = {}
MISSING = {"orjson", "msgspec", "zmq", "pybase64", "IPython", "uvloop", "setproctitle", "xgrammar",
           "sgl_kernel", "flashinfer", "aiter", "vllm"}
REAL = {
    "sglang.srt.mem_cache.radix_cache",
    "sglang.srt.mem_cache.base_prefix_cache",
    "sglang.srt.mem_cache.allocator.paged",
    "sglang.srt.mem_cache.allocator.token",
    "sglang.srt.mem_cache.allocation",
    "sglang.srt.layers.attention.torch_native_backend",
    "sglang.srt.layers.moe.fused_moe_native",
    "sglang.srt.layers.moe.topk",
    "sglang.srt.layers.layernorm",
    "sglang.srt.layers.rotary_embedding.base",
    "sglang.srt.layers.rotary_embedding.utils",
    "sglang.srt.layers.rotary_embedding.rope_variant",
    "sglang.srt.layers.sampler",
    "sglang.srt.model_executor.forward_batch_info",
    "sglang.srt.models.llava",
    "sglang.srt.multimodal.mm_utils",
}
FAILED = []
NOT_FOUND = []            # sglang modules asked for that do not exist under REF (a staged copy that misses a file): stubbed
ABSENT_ATTRS = set()      # names a stubbed module must NOT pretend to have (ref_model.py: "EntryClass", which the model registry probes)
SUBMODULE_ATTRS = False   # ref_model.py: see _Stub.__getattr__
TRY_PACKAGES = False      # ref_objects.py: run package __init__ files too (degrading to a stub when one cannot import)


class _StubLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        pass


class _PkgLoader(importlib.abc.Loader):
    def __init__(self, paths):
        self.paths = paths

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = self.paths
        return m

    def exec_module(self, m):
        pass


class _WrapLoader(importlib.abc.Loader):
    def __init__(self, inner):
        self.inner = inner

    def create_module(self, spec):
        return self.inner.create_module(spec)

    def exec_module(self, m):
        try:
            self.inner.exec_module(m)
        except Exception as e:  # degrade to a stub
            FAILED.append((m.__name__, f"{type(e).__name__}: {str(e)[:100]}"))
            m.__class__ = _Stub
            if not hasattr(m, "__path__"):
                m.__path__ = []


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path, target=None):
        root = name.split(".")[0]
        if root == "msgspec" and MSGSPEC_EMULATION:
            return importlib.machinery.ModuleSpec(name, _MsgspecLoader(), is_package=True)
        if root in MISSING:
            return importlib.machinery.ModuleSpec(name, _StubLoader(), is_package=True)
        if root == "sglang":
            spec = importlib.machinery.PathFinder.find_spec(name, path)
            if spec is None:
                NOT_FOUND.append(name)
                return importlib.machinery.ModuleSpec(name, _StubLoader(), is_package=True)
            if name in REAL:
                return spec
            if spec.submodule_search_locations is not None and not (TRY_PACKAGES and name != "sglang" and spec.loader is not None):
                return importlib.machinery.ModuleSpec(
                    name, _PkgLoader(list(spec.submodule_search_locations)), is_package=True)
            spec.loader = _WrapLoader(spec.loader)
            return spec
        return None


def install_hook():
    sys.path.insert(0, str(REF))
    sys.meta_path.insert(0, _Finder())


def ref(mod: str):
    return importlib.import_module(mod)


# ----------------------------------------------------------------------------- generators
def gen_attention():
    tnb = ref("sglang.srt.layers.attention.torch_native_backend").TorchNativeAttnBackend
    g = torch.Generator().manual_seed(1234)
    cases = {}
    for name, (Hq, Hkv, D) in {"gqa_d64": (4, 2, 64), "mha_d128": (2, 2, 128), "gqa4_d128": (8, 2, 128)}.items():
        slots, max_ctx = 96, 40
        prefix = torch.tensor([0, 5, 17, 8])
        extend = torch.tensor([9, 7, 3, 1])
        seq = prefix + extend
        B = len(seq)
        perm = torch.randperm(slots - 1, generator=g) + 1
        req_to_token = torch.zeros((B + 1, max_ctx), dtype=torch.int32)
        req_pool = torch.tensor([2, 4, 1, 3])
        off = 0
        for i in range(B):
            req_to_token[req_pool[i], : seq[i]] = perm[off: off + seq[i]].to(torch.int32)
            off += int(seq[i])
        k_cache = (torch.randn((slots, Hkv, D), generator=g) * 0.5).to(torch.bfloat16)
        v_cache = (torch.randn((slots, Hkv, D), generator=g) * 0.5).to(torch.bfloat16)
        T = int(extend.sum())
        q = (torch.randn((T, Hq, D), generator=g) * 0.5).to(torch.bfloat16)
        scaling = D ** -0.5
        o = torch.empty_like(q)
        tnb._run_sdpa_forward_extend(None, q, o, k_cache, v_cache, req_to_token, req_pool, seq, prefix, extend,
                                     scaling=scaling, enable_gqa=Hq != Hkv, causal=True)
        qd = (torch.randn((B, Hq, D), generator=g) * 0.5).to(torch.bfloat16)
        od = torch.empty_like(qd)
        tnb._run_sdpa_forward_decode(None, qd, od, k_cache, v_cache, req_to_token, req_pool, seq,
                                     scaling=scaling, enable_gqa=Hq != Hkv, causal=False)
        cases[name] = dict(q=q, out_extend=o, q_decode=qd, out_decode=od, k_cache=k_cache, v_cache=v_cache,
                           req_to_token=req_to_token, req_pool_indices=req_pool, seq_lens=seq,
                           extend_prefix_lens=prefix, extend_seq_lens=extend, scaling=scaling)
        # sliding-window layers (torch_native_backend.py:36-48,150-156,251-257): the same inputs with a window of 6
        W = 6
        ow, odw = torch.empty_like(q), torch.empty_like(qd)
        backend = types.SimpleNamespace(_make_sliding_window_mask=tnb._make_sliding_window_mask)
        tnb._run_sdpa_forward_extend(backend, q, ow, k_cache, v_cache, req_to_token, req_pool, seq, prefix, extend,
                                     scaling=scaling, enable_gqa=Hq != Hkv, causal=True, sliding_window_size=W)
        tnb._run_sdpa_forward_decode(backend, qd, odw, k_cache, v_cache, req_to_token, req_pool, seq,
                                     scaling=scaling, enable_gqa=Hq != Hkv, causal=False, sliding_window_size=W)
        cases[name].update(sliding_window=W, out_extend_window=ow, out_decode_window=odw)
    torch.save(cases, OUT / "attention_torch_native.pt")


def gen_elementwise():
    ln = ref("sglang.srt.layers.layernorm")
    rb = ref("sglang.srt.layers.rotary_embedding.base")
    ru = ref("sglang.srt.layers.rotary_embedding.utils")
    rv = ref("sglang.srt.layers.rotary_embedding.rope_variant")
    g = torch.Generator().manual_seed(7)
    out = {}
    # RMSNorm.forward_native (layernorm.py:777-826) with a minimal fake self
    for name, (T, H) in {"t7_h2048": (7, 2048), "t3_h896": (3, 896)}.items():
        w = (1.0 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16)
        x = torch.randn((T, H), generator=g).to(torch.bfloat16)
        r = torch.randn((T, H), generator=g).to(torch.bfloat16)
        fake = types.SimpleNamespace(weight=w, variance_epsilon=1e-5, hidden_size=H, override_orig_dtype=None,
                                     fp32_residual=False, variance_size_override=None, cast_x_before_out_mul=False)
        y = ln.RMSNorm.forward_native(fake, x.clone())
        y2, r2 = ln.RMSNorm.forward_native(fake, x.clone(), r.clone())
        out[f"rmsnorm_{name}"] = dict(x=x, residual=r, weight=w, eps=1e-5, out=y, out_fused=y2, residual_out=r2)
    # RotaryEmbedding.forward_native (base.py:236-276) + Llama3 inv_freq (rope_variant.py:560-580)
    for name, (Hq, Hk, D, neox, llama3) in {"neox_d128_llama3": (4, 2, 128, True, True),
                                            "neox_d64": (14, 2, 64, True, False),
                                            "gptj_d64": (2, 2, 64, False, False)}.items():
        base, max_pos = 500000.0, 512
        fake = types.SimpleNamespace(rotary_dim=D, base=base, _force_native=False, max_position_embeddings=max_pos,
                                     scaling_factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                     orig_max_position=8192)
        if llama3:
            # Llama3RotaryEmbedding._compute_inv_freq calls super()._compute_inv_freq(base)
            class _L3(rv.Llama3RotaryEmbedding):
                def __init__(self):  # skip nn.Module / platform init
                    pass
            l3 = _L3.__new__(_L3)
            l3.__dict__.update(fake.__dict__)
            inv = rv.Llama3RotaryEmbedding._compute_inv_freq(l3, base)
        else:
            inv = rb.RotaryEmbedding._compute_inv_freq(fake, base)
        fake._compute_inv_freq = lambda b, inv=inv: inv
        cache_f32 = rb.RotaryEmbedding._compute_cos_sin_cache(fake)
        cache = cache_f32.to(torch.bfloat16)  # HIP path casts the cache to the model dtype (base.py:104-106)
        T = 6
        pos = torch.tensor([0, 1, 5, 100, 257, 511])
        q = torch.randn((T, Hq * D), generator=g).to(torch.bfloat16)
        k = torch.randn((T, Hk * D), generator=g).to(torch.bfloat16)
        fwd = types.SimpleNamespace(cos_sin_cache=cache, head_size=D, rotary_dim=D, is_neox_style=neox,
                                    _apply_rotary_emb_wrapped=ru.apply_rotary_emb)
        qo, ko = rb.RotaryEmbedding.forward_native(fwd, pos, q.clone(), k.clone())
        out[f"rope_{name}"] = dict(positions=pos, q=q, k=k, q_out=qo, k_out=ko, cache=cache, cache_f32=cache_f32,
                                   inv_freq=inv, head_size=D, is_neox=neox, base=base, max_pos=max_pos,
                                   llama3=llama3)
    torch.save(out, OUT / "elementwise_native.pt")


def gen_moe():
    topk_mod = ref("sglang.srt.layers.moe.topk")
    native = ref("sglang.srt.layers.moe.fused_moe_native")
    g = torch.Generator().manual_seed(11)
    M, K, N, E, k = 9, 64, 48, 8, 2
    x = (torch.randn((M, K), generator=g) * 0.5).to(torch.bfloat16)
    w13 = (torch.randn((E, 2 * N, K), generator=g) * 0.1).to(torch.bfloat16)
    w2 = (torch.randn((E, K, N), generator=g) * 0.1).to(torch.bfloat16)
    logits = torch.randn((M, E), generator=g).to(torch.bfloat16)
    tw, ti = topk_mod.fused_topk_torch_native(x, logits, k, True)
    layer = types.SimpleNamespace(w13_weight=w13, w2_weight=w2, num_experts=E,
                                  moe_runner_config=types.SimpleNamespace(apply_router_weight_on_input=False,
                                                                          activation="silu", gemm1_alpha=None,
                                                                          gemm1_clamp_limit=None))
    disp = types.SimpleNamespace(hidden_states=x, topk_output=(tw, ti, logits))

    class _Combine:
        def __init__(self, hidden_states):
            self.hidden_states = hidden_states
    with mock.patch.object(native, "StandardCombineInput", _Combine):
        y = native.fused_moe_forward_native(layer, disp).hidden_states
    torch.save(dict(x=x, w13=w13, w2=w2, router_logits=logits, topk=k, topk_weights=tw, topk_ids=ti, out_einsum=y),
               OUT / "moe_native.pt")


def gen_sampler():
    s = ref("sglang.srt.layers.sampler")
    g = torch.Generator().manual_seed(5)
    B, V = 6, 1000
    logits = torch.randn((B, V), generator=g) * 3
    probs = torch.softmax(logits, dim=-1)
    top_ks = torch.tensor([50, 1, 1 << 30, 20, 1 << 30, 5], dtype=torch.int32)
    top_ps = torch.tensor([0.9, 1.0, 0.5, 1.0, 1.0, 0.3])
    min_ps = torch.tensor([0.0, 0.0, 0.0, 0.05, 0.02, 0.0])
    captured = {}

    def fake_multinomial(p, num_samples):
        captured["kept"] = p.clone()
        return torch.zeros((p.shape[0], 1), dtype=torch.int64)
    res = {}
    for need_min_p in (False, True):
        with mock.patch.object(torch, "multinomial", fake_multinomial):
            ids0 = s.top_k_top_p_min_p_sampling_from_probs_torch(probs.clone(), top_ks, top_ps, min_ps, need_min_p,
                                                                 None, torch.zeros(B, dtype=torch.int64))
        res[f"kept_sorted_minp{int(need_min_p)}"] = captured["kept"]
        res[f"rank0_ids_minp{int(need_min_p)}"] = ids0
    res.update(probs=probs, top_ks=top_ks, top_ps=top_ps, min_ps=min_ps)
    torch.save(res, OUT / "sampler_torch.pt")


def gen_radix(seed_base=100, steps=220, pages=(1, 4), vocab=6, out=None):
    """`out=None`: the committed fixture (seed 100, 220 steps, pages 1 / 4).  tests/test_radix_cache.py::test_live_differential_traces
    calls this with other seeds / lengths / page sizes / vocabularies (a fresh process, an output path) and replays the result on
    the product's tree: the comparison is then against the reference's live behaviour, not one recorded trace."""
    rc = ref("sglang.srt.mem_cache.radix_cache")
    bpc = ref("sglang.srt.mem_cache.base_prefix_cache")
    RadixCache, RadixKey = rc.RadixCache, rc.RadixKey
    traces = {}

    def key(ids):
        return RadixKey(token_ids=array("q", ids))

    for page_size in pages:
        rnd = random.Random(seed_base + page_size)
        alloc = mock.Mock()
        alloc.device = "cpu"
        tree = RadixCache.create_simulated(mock_allocator=alloc, page_size=page_size)
        ops = []
        next_slot = 1
        held = []  # nodes we locked
        for step in range(steps):
            r = rnd.random()
            if r < 0.45:
                n = rnd.randint(1, 14)
                ids = [rnd.randrange(vocab) for _ in range(n)]
                vals = list(range(next_slot, next_slot + n))
                next_slot += n
                res = tree.insert(bpc.InsertParams(key=key(ids), value=torch.tensor(vals, dtype=torch.int64)))
                ops.append(dict(op="insert", ids=ids, vals=vals, prefix_len=int(res.prefix_len)))
            elif r < 0.8:
                n = rnd.randint(0, 14)
                ids = [rnd.randrange(vocab) for _ in range(n)]
                m = tree.match_prefix(bpc.MatchPrefixParams(key=key(ids)))
                if rnd.random() < 0.3 and m.last_device_node is not tree.root_node:
                    tree.inc_lock_ref(m.last_device_node)
                    held.append(m.last_device_node)
                    locked = True
                else:
                    locked = False
                ops.append(dict(op="match", ids=ids, indices=m.device_indices.tolist(), lock=locked))
            elif r < 0.9 and held:
                idx = rnd.randrange(len(held))
                node = held.pop(idx)
                tree.dec_lock_ref(node)
                ops.append(dict(op="unlock", which=idx))
            else:
                n = rnd.randint(1, 10)
                alloc.reset_mock()
                res = tree.evict(bpc.EvictParams(num_tokens=n))
                freed = []
                for c in alloc.free_segment.call_args_list:
                    freed.append(c.args[0].tolist())
                ops.append(dict(op="evict", num_tokens=n, num_evicted=int(res.num_tokens_evicted), freed=freed))
            ops[-1].update(evictable=int(tree.evictable_size()), protected=int(tree.protected_size()),
                           total=int(tree.total_size()))
        traces[f"page{page_size}"] = ops

    # the reference's own __main__ scenario (radix_cache.py:849-863)
    tree = RadixCache.create_simulated()
    for ids in ([1, 2, 3], [1, 2, 3], [1, 2, 4, 5], [1, 2, 4, 5, 6, 7], [8, 9, 10, 11, 12]):
        tree.insert(bpc.InsertParams(key=key(ids)))
    m = tree.match_prefix(bpc.MatchPrefixParams(key=key([1, 2, 3, 13, 14])))
    traces["main_scenario"] = dict(match=m.device_indices.tolist(), total=int(tree.total_size()))
    Path(out or OUT / "radix_trace.json").write_text(json.dumps(traces))


def gen_host_int():
    paged = ref("sglang.srt.mem_cache.allocator.paged")
    alloc_mod = ref("sglang.srt.mem_cache.allocation")
    fbi = ref("sglang.srt.model_executor.forward_batch_info")
    rnd = random.Random(9)
    cases = []
    for page_size in (1, 4, 16):
        for _ in range(6):
            bs = rnd.randint(1, 7)
            prefix = [rnd.randint(0, 40) for _ in range(bs)]
            ext = [rnd.randint(1, 50) for _ in range(bs)]
            seq = [p + e for p, e in zip(prefix, ext)]
            # a consistent last_loc: last slot of the prefix's (possibly partial) page
            last_loc, used_pages = [], 0
            for p in prefix:
                if p == 0:
                    last_loc.append(-1)
                else:
                    page = 1000 + used_pages
                    used_pages += 1
                    last_loc.append(page * page_size + (p - 1) % page_size)
            n_free = sum((s + page_size - 1) // page_size for s in seq) + 3
            free_pages = rnd.sample(range(1, 900), n_free)
            out = torch.full((sum(ext),), -7, dtype=torch.int64)
            paged.alloc_extend_naive(torch.tensor(prefix), torch.tensor(seq), torch.tensor(last_loc),
                                     torch.tensor(free_pages), out, page_size, "cpu")
            cases.append(dict(page_size=page_size, prefix=prefix, seq=seq, last_loc=last_loc, free_pages=free_pages,
                              out=out.tolist()))
    pos_cases = []
    for _ in range(5):
        bs = rnd.randint(1, 6)
        prefix = torch.tensor([rnd.randint(0, 30) for _ in range(bs)])
        ext = torch.tensor([rnd.randint(1, 9) for _ in range(bs)])
        p, s = fbi.compute_position_torch(prefix, ext)
        pos_cases.append(dict(prefix=prefix.tolist(), extend=ext.tolist(), positions=p.tolist(), start=s.tolist()))
    r2t = torch.arange(5 * 11, dtype=torch.int32).reshape(5, 11)
    rp = torch.tensor([3, 0, 4])
    pl = torch.tensor([0, 4, 11])
    ll = alloc_mod.get_last_loc_torch(r2t, rp, pl)
    clamp = fbi._clamp_position_native(torch.tensor([0, 1, 5, 9]))
    (OUT / "host_int.json").write_text(json.dumps(dict(
        alloc_extend=cases, compute_position=pos_cases,
        get_last_loc=dict(req_pool=rp.tolist(), prefix=pl.tolist(), out=ll.tolist()),
        clamp_position=dict(seq=[0, 1, 5, 9], out=clamp.tolist()))))


def gen_llava_anyres():
    """LLaVA-1.6 anyres: the reference's own pad_input_ids (models/llava.py:79-143) and the feature packing + embedding
    substitution of LlavaBaseForCausalLM.forward (:168-460: anyres grid, spatial unpad, image_newline column, base tile
    first, then the copy into the extend range) driven with a stand-in `self` -- vision tower and language model are
    replaced by recorders, every line in between is the reference's."""
    import numpy as np

    lv = ref("sglang.srt.models.llava")
    mm = ref("sglang.srt.multimodal.mm_utils")
    if not callable(getattr(lv.flatten_nested_list, "__code__", None) and lv.flatten_nested_list):
        # sglang.srt.utils does not import here (its third-party dependencies are absent), so llava.py holds a stub for
        # this one list helper; it is executed from the reference's own source text (utils/common.py flatten_nested_list)
        import ast

        src = (REF / "sglang/srt/utils/common.py").read_text()
        fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "flatten_nested_list")
        ns = {}
        exec(compile(ast.Module(body=[fn], type_ignores=[]), "utils/common.py", "exec"), ns)
        lv.flatten_nested_list = ns["flatten_nested_list"]
    g = torch.Generator().manual_seed(16)
    S, P, HID, VOCAB, IMG = 336, 14, 8, 100, 90
    pinpoints = [[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]]            # llava-v1.6 config.json
    side = S // P
    flen = side * side
    newline = torch.randn(HID, generator=g).to(torch.bfloat16)
    table = torch.randn((VOCAB, HID), generator=g).to(torch.bfloat16)
    out = dict(image_size=S, patch_size=P, hidden=HID, vocab=VOCAB, image_token_index=IMG, pinpoints=pinpoints,
               image_newline=newline, embed_table=table, cases=[])
    grid_cases = []
    for wh in [(1000, 600), (600, 1000), (640, 480), (336, 336), (2000, 500), (300, 1200), (1024, 1024), (500, 333)]:
        gw, gh = mm.get_anyres_image_grid_shape(wh, pinpoints, S)
        grid_cases.append(dict(size=wh, grid=(gw, gh), unpad=mm.unpad_image_shape(gh * side, gw * side, wh)))
    out["grid_cases"] = grid_cases

    captured = {}

    class FakeLM:
        def __call__(self, input_ids, positions, forward_batch, input_embeds=None):
            captured["embeds"] = input_embeds.clone()
            return "lm-out"

    fake_lm = FakeLM()
    fake_lm.model = types.SimpleNamespace(embed_tokens=lambda ids: torch.nn.functional.embedding(ids, table).clone(), image_newline=newline)
    for wh, prefix_len in [((1000, 600), 0), ((600, 1000), 0), ((640, 480), 7), ((336, 336), 0), ((2000, 500), 1500)]:
        gw, gh = mm.get_anyres_image_grid_shape(wh, pinpoints, S)
        tiles = 1 + gw * gh
        feats = torch.randn((tiles, flen, HID), generator=g).to(torch.bfloat16)

        fake = types.SimpleNamespace(
            config=types.SimpleNamespace(vocab_size=VOCAB, image_grid_pinpoints=pinpoints, image_aspect_ratio="anyres",
                                         image_token_index=IMG),
            image_grid_pinpoints=pinpoints, image_size=S, patch_size=P, image_feature_len=flen, num_patches_per_side=side,
            mm_patch_merge_type="spatial_unpad", vision_tower=types.SimpleNamespace(device="cpu", config=types.SimpleNamespace(image_size=S)),
            language_model=fake_lm, encode_images=lambda pix, feats=feats: feats, _infer_image_aspect_ratio=lambda items: "anyres")
        item = types.SimpleNamespace(feature=np.zeros((tiles, 3, 2, 2), dtype=np.float32), image_sizes=[wh], pad_value=1_000_123,
                                     modality=lv.Modality.IMAGE)
        inputs = types.SimpleNamespace(mm_items=[item], image_offsets=None, image_pad_len=None)
        ids = array("q", [1, 2, 3, 4, 5, 6, 7, 8, 9, IMG, 11, 12, 13])
        padded = lv.LlavaBaseForCausalLM.pad_input_ids(fake, ids, inputs)
        total = len(padded)
        ext = total - prefix_len
        fb = types.SimpleNamespace(mm_inputs=[inputs], forward_mode=types.SimpleNamespace(is_extend=lambda: True, is_decode=lambda: False),
                                   extend_start_loc=torch.tensor([0]), extend_seq_lens=torch.tensor([ext]),
                                   extend_prefix_lens_cpu=[prefix_len], batch_size=1)
        in_ids = torch.tensor(list(padded[prefix_len:]), dtype=torch.int64)
        positions = torch.arange(prefix_len, total)
        lv.LlavaBaseForCausalLM.forward(fake, in_ids.clone(), positions, fb)
        out["cases"].append(dict(size=wh, grid=(gw, gh), tiles=tiles, tile_features=feats, prompt=list(ids), padded=list(padded),
                                 offsets=list(inputs.image_offsets), pad_len=list(inputs.image_pad_len), prefix_len=prefix_len,
                                 input_embeds=captured["embeds"]))
    torch.save(out, OUT / "llava_anyres.pt")


def main():
    if not REF.exists():
        raise SystemExit("/root/reference not present: goldens can only be regenerated in the build container")
    install_hook()
    if len(sys.argv) > 1 and sys.argv[1] == "--radix-live":            # --radix-live SEED STEPS VOCAB OUT: a fresh trace, pages 1 / 4 / 16
        gen_radix(seed_base=int(sys.argv[2]), steps=int(sys.argv[3]), pages=(1, 4, 16), vocab=int(sys.argv[4]), out=sys.argv[5])
        return
    torch.manual_seed(0)
    for fn in (gen_attention, gen_elementwise, gen_moe, gen_sampler, gen_radix, gen_host_int, gen_llava_anyres):
        fn()
        print("ok", fn.__name__)
    print("degraded-to-stub reference modules:", len(FAILED))


if __name__ == "__main__":
    main()
