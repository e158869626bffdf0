"""The drop-in boundary against the reference's own sources.

tests/golden/reference_contract.json is the ast-extracted contract of the REAL reference (signatures, dataclass
fields, class constants of every module the plugin touches; tests/golden/gen_contract.py).  This file
  1. checks that every operator of sglang_amd/layers is call-compatible with the reference operator it stands for
     (same parameter names in the same order, a default wherever the reference has one);
  2. builds a stand-in `sglang` package from the contract -- every module path, function and method exists with the
     reference's exact signature, registries behave as the cited code does -- and EXECUTES plugin.load() and the
     platform entry point against it, then drives the registered objects the way the reference would:
     attention factory with a reference-shaped ModelRunner, RMSNorm / TopK forwards bound to reference-shaped op
     instances, the sampler factory, the fused-MoE slot with quantised / biased inputs.
No GPU is needed: the HIP library loads on the CPU (symbols only), tensors stay on the CPU so the operators take
their documented delegation paths.
"""
import importlib
import inspect
import json
import sys
import types
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
CONTRACT = json.loads((ROOT / "tests" / "golden" / "reference_contract.json").read_text())
MODS = CONTRACT["modules"]


def ref(module, name):
    return MODS[module]["names"][name]


def ref_params(module, name, method=None, drop_self=True):
    rec = ref(module, name)
    if method is not None:
        rec = rec["methods"][method]
    ps = rec["params"]
    if drop_self and ps and ps[0]["name"] in ("self", "cls"):
        ps = ps[1:]
    return ps


def assert_call_compatible(ours, ref_ps, what):
    """Every way the reference can be called must bind on ours: same names, same order for positionals, a default
    wherever the reference has one; ours may add trailing optional parameters."""
    sig = inspect.signature(ours)
    mine = [p for p in sig.parameters.values() if p.name not in ("self", "cls")]
    names = [p.name for p in mine]
    pos_ref = [p for p in ref_ps if p["kind"] == "pos"]
    for i, p in enumerate(pos_ref):
        assert i < len(mine) and mine[i].name == p["name"], f"{what}: positional #{i} is '{names[i] if i < len(names) else None}', reference has '{p['name']}'"
        if p["default"]:
            assert mine[i].default is not inspect.Parameter.empty, f"{what}: '{p['name']}' needs a default"
    for p in ref_ps:
        if p["kind"] == "kw":
            assert p["name"] in sig.parameters, f"{what}: keyword '{p['name']}' missing"
    for p in mine[len(pos_ref):]:
        if p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD):
            continue
        assert p.default is not inspect.Parameter.empty or p.kind == p.KEYWORD_ONLY and p.name in [q["name"] for q in ref_ps], \
            f"{what}: extra parameter '{p.name}' must be optional"


# ------------------------------------------------------------------------------------------ 1. signatures
def test_operator_signatures_match_the_reference():
    from sglang_amd.layers import activation, layernorm, rotary_embedding, sampler
    from sglang_amd.layers.attention.base_attn_backend import AttentionBackend
    from sglang_amd.layers.attention.hip_backend import HipAttnBackend
    from sglang_amd.layers.moe import topk

    for fwd in ("forward_cuda", "forward_native"):
        assert_call_compatible(layernorm.RMSNorm.forward, ref_params("sglang.srt.layers.layernorm", "RMSNorm", fwd), f"RMSNorm vs {fwd}")
        assert_call_compatible(rotary_embedding.RotaryEmbedding.forward,
                               ref_params("sglang.srt.layers.rotary_embedding.base", "RotaryEmbedding", fwd), f"RotaryEmbedding vs {fwd}")
        assert_call_compatible(topk.TopK.forward, ref_params("sglang.srt.layers.moe.topk", "TopK", fwd), f"TopK vs {fwd}")
    assert_call_compatible(activation.SiluAndMul.forward, ref_params("sglang.srt.layers.activation", "SiluAndMul", "forward_native"), "SiluAndMul")
    assert_call_compatible(sampler.Sampler.forward, ref_params("sglang.srt.layers.sampler", "Sampler", "forward"), "Sampler.forward")
    for m in ("init_forward_metadata", "init_forward_metadata_out_graph", "init_forward_metadata_in_graph", "init_cuda_graph_state",
              "get_cuda_graph_seq_len_fill_value", "forward", "forward_decode", "forward_extend", "forward_mixed", "support_triton"):
        assert_call_compatible(getattr(HipAttnBackend, m), ref_params("sglang.srt.layers.attention.base_attn_backend", "AttentionBackend", m),
                               f"HipAttnBackend.{m}")
    # the class-level switches the graph runner / scheduler read
    attrs = ref("sglang.srt.layers.attention.base_attn_backend", "AttentionBackend")["attrs"]
    for a in ("needs_cpu_seq_lens", "extend_dummy_seqs_capped_by_req_pool"):
        assert a in attrs and hasattr(HipAttnBackend, a), a
    assert issubclass(HipAttnBackend, AttentionBackend)


def test_functional_namespace_matches_sgl_kernel():
    """kernels.py mirrors of sgl_kernel.{rmsnorm, fused_add_rmsnorm, silu_and_mul, rotary_embedding, topk_softmax, ...}:
    the reference's positional call forms bind (enable_pdl is CUDA-only and has no counterpart)."""
    from sglang_amd import kernels

    ns = {}
    for f in CONTRACT["sgl_kernel"].values():
        ns.update(f)
    pairs = {"rmsnorm": ("x", ["input", "weight", "eps", "out"]), "fused_add_rmsnorm": ("x", ["input", "residual", "weight", "eps"]),
             "silu_and_mul": ("x", ["input", "out"]),
             "rotary_embedding": ("positions", ["positions", "query", "key", "head_size", "cos_sin_cache", "is_neox"])}
    for name, (first, ref_order) in pairs.items():
        got = [p["name"] for p in ns[name]["params"] if p["name"] != "enable_pdl"]
        assert got == ref_order, (name, got)
        mine = list(inspect.signature(getattr(kernels, name)).parameters)
        assert len(mine) >= len(ref_order) and mine[0] == first and mine[1:len(ref_order)] == ref_order[1:], (name, mine)
    assert ns["top_k_renorm_prob"]["value"] == "top_k_renorm_probs" and ns["top_p_renorm_prob"]["value"] == "top_p_renorm_probs"
    assert [p["name"] for p in ns["top_k_renorm_probs"]["params"]][:2] == ["probs", "top_k"]
    assert [p["name"] for p in ns["top_p_renorm_probs"]["params"]][:2] == ["probs", "top_p"]
    assert list(inspect.signature(kernels.top_k_renorm_prob).parameters) == ["probs", "top_k"]
    assert list(inspect.signature(kernels.top_p_renorm_prob).parameters) == ["probs", "top_p"]


def test_entry_points_are_declared_as_the_reference_discovers_them():
    import tomli

    ep = tomli.loads((ROOT / "pyproject.toml").read_text())["project"]["entry-points"]
    assert ref("sglang.srt.plugins", "GENERAL_PLUGINS_GROUP")["value"] in ep
    assert ref("sglang.srt.plugins", "PLATFORM_PLUGINS_GROUP")["value"] in ep
    mod, fn = ep["sglang.srt.platforms"]["hip_mi355x"].split(":")
    assert callable(getattr(importlib.import_module(mod), fn))
    mod, fn = ep["sglang.srt.plugins"]["sglang_amd"].split(":")
    assert callable(getattr(importlib.import_module(mod), fn))
    from sglang_amd import platform

    assert platform.activate() is None or torch.cuda.is_available()      # no gfx950 device -> "hardware not available"
    # every factory the platform overrides exists on the reference interface with a compatible signature
    methods = ref("sglang.srt.platforms.interface", "SRTPlatform")["methods"]
    for m in ("get_dispatch_key_name", "get_default_attention_backend", "support_cuda_graph", "supports_fp8", "get_graph_runner_cls",
              "get_mha_kv_pool_cls", "get_paged_allocator_cls", "init_backend"):
        assert m in methods and [p["name"] for p in methods[m]["params"]] == ["self"], m
    assert "is_out_of_tree" in ref("sglang.srt.platforms.device_mixin", "DeviceMixin")["methods"]
    assert "OOT" in ref("sglang.srt.platforms.device_mixin", "PlatformEnum")["attrs"]


# ------------------------------------------------------------------------------------------ 2. stand-in package
def _fn_src(name, params, body="raise NotImplementedError('stand-in')", deco=""):
    parts, seen_kw = [], False
    for p in params:
        if p["kind"] == "var":
            parts.append("*" + p["name"]); seen_kw = True
        elif p["kind"] == "varkw":
            parts.append("**" + p["name"])
        else:
            if p["kind"] == "kw" and not seen_kw:
                parts.append("*"); seen_kw = True
            parts.append(p["name"] + ("=None" if p["default"] else ""))
    return f"{deco}def {name}({', '.join(parts)}):\n    {body}\n"


@pytest.fixture
def fake_sglang(monkeypatch):
    """A `sglang` package whose modules, names and signatures are the contract's; behaviour is filled in only where
    the cited reference code has behaviour the plugin depends on (registries, dispatch rules)."""
    created = {}

    def module(name):
        if name in created:
            return created[name]
        m = types.ModuleType(name)
        m.__path__ = []
        created[name] = m
        monkeypatch.setitem(sys.modules, name, m)
        if "." in name:
            parent, _, child = name.rpartition(".")
            setattr(module(parent), child, m)
        return m

    for modname, rec in MODS.items():
        m = module(modname)
        for name, r in rec["names"].items():
            if "." in name:
                continue
            if r["kind"] == "constant":
                setattr(m, name, r["value"])
            elif r["kind"] == "function":
                ns = {}
                exec(_fn_src(name, r["params"]), ns)
                setattr(m, name, ns[name])
            elif r["kind"] == "class":
                body = {}
                for meth, mr in r["methods"].items():
                    ns = {}
                    deco = "".join(f"@{d}\n" for d in mr["decorators"] if d in ("classmethod", "staticmethod"))
                    exec(_fn_src(meth, mr["params"], deco=deco), ns)
                    body[meth] = ns[meth]
                for a, v in r["attrs"].items():
                    body.setdefault(a, v)
                body["__contract_fields__"] = [f["name"] for f in r["fields"]]
                setattr(m, name, type(name, (), body))

    # ---- behaviour, restated from the cited lines -----------------------------------------------------
    import enum

    dm = created["sglang.srt.platforms.device_mixin"]
    PlatformEnum = enum.Enum("PlatformEnum", list(ref("sglang.srt.platforms.device_mixin", "PlatformEnum")["attrs"]))
    dm.PlatformEnum = PlatformEnum

    class DeviceMixin:                                                    # device_mixin.py:101-145
        _enum = PlatformEnum.UNSPECIFIED

        def is_out_of_tree(self):
            return self._enum == PlatformEnum.OOT

    dm.DeviceMixin = DeviceMixin
    created["sglang.srt.platforms.cuda"].CudaDeviceMixin = type("CudaDeviceMixin", (DeviceMixin,), {"_enum": PlatformEnum.CUDA})
    iface = created["sglang.srt.platforms.interface"]
    iface.SRTPlatform = type("SRTPlatform", (DeviceMixin,), {k: v for k, v in vars(iface.SRTPlatform).items() if not k.startswith("__")})
    iface.SRTPlatform.get_dispatch_key_name = lambda self: "native"       # interface.py:133-142

    reg = created["sglang.srt.layers.attention.attention_registry"]
    reg.ATTENTION_BACKENDS = {}

    def register_attention_backend(name):                                 # attention_registry.py:34-39
        def deco(fn):
            reg.ATTENTION_BACKENDS[name] = fn
            return fn
        return deco

    reg.register_attention_backend = register_attention_backend
    sa = created["sglang.srt.server_args"]
    sa.ATTENTION_BACKEND_CHOICES = ["triton", "torch_native"]
    sa.add_attention_backend_choices = lambda choices: sa.ATTENTION_BACKEND_CHOICES.extend(choices)   # server_args.py:416-417

    smp = created["sglang.srt.layers.sampler"]
    smp._SAMPLER_FACTORIES = {}

    class RefSampler(torch.nn.Module):                                    # sampler.py:71-97
        def __init__(self):
            super().__init__()
            self.tp_sync_group = None
            self.synced = 0
            self.output_logprob_processor = None

        def _preprocess_logits(self, logits, sampling_info):
            self.preprocessed = True
            return logits

        def _sync_token_ids_across_tp(self, batch_next_token_ids, sampling_info):
            self.synced += 1

        def forward(self, *a, **k):            # reached only for the calls the gfx950 forward declines (plugin.sampler_declines)
            self.reference_forwards = getattr(self, "reference_forwards", 0) + 1
            return "the reference's forward"

    smp.Sampler = RefSampler
    smp.register_sampler_backend = lambda backend, factory: smp._SAMPLER_FACTORIES.__setitem__(backend, factory)   # sampler.py:531-542

    base = created["sglang.srt.layers.moe.moe_runner.base"]

    class FusedOpPool:                                                     # base.py:115-143
        _fused_funcs = {}

        @classmethod
        def register_fused_func(cls, a2a_backend_name, runner_backend_name, fused_func):
            key = (a2a_backend_name, runner_backend_name)
            if key in cls._fused_funcs:
                raise ValueError(f"Fused function for {a2a_backend_name} to {runner_backend_name} is already registered.")
            cls._fused_funcs[key] = fused_func

        @classmethod
        def get_fused_func(cls, dispatch_name, runner_name):
            return cls._fused_funcs.get((dispatch_name, runner_name))

    base.FusedOpPool = FusedOpPool

    def register_fused_func(a2a_backend_name, runner_backend_name):
        def decorator(fused_func):
            FusedOpPool.register_fused_func(a2a_backend_name, runner_backend_name, fused_func)
            return fused_func
        return decorator

    base.register_fused_func = register_fused_func
    tri = created["sglang.srt.layers.moe.moe_runner.triton"]
    tri.calls = []

    @register_fused_func("none", "triton")                                # triton.py:180 registers at import
    def fused_experts_none_to_triton(dispatch_output, quant_info, runner_config):
        tri.calls.append("reference")
        return "reference-result"

    tri.fused_experts_none_to_triton = fused_experts_none_to_triton
    std = created["sglang.srt.layers.moe.token_dispatcher.standard"]
    std.StandardCombineInput = lambda hidden_states: ("combine", hidden_states)

    fo = created["sglang.kernels.fused_op"]
    current = {"platform": None}
    plat_pkg = created["sglang.srt.platforms"]

    class BaseFusedOp(torch.nn.Module):                                   # fused_op.py:332-391, 533-562
        _oot_forward_registry = {}

        @classmethod
        def register_oot_forward(cls, op_cls, fn, platform_key):
            cls._oot_forward_registry.setdefault(platform_key, {})[op_cls] = fn

        def _resolve_forward_method(self):
            p = current["platform"]
            if p is not None and p.is_out_of_tree():
                registered = self._oot_forward_registry.get(p.get_dispatch_key_name(), {}).get(type(self))
                if registered is not None:
                    return registered.__get__(self)
            return self.forward_native

        def forward(self, *args, **kwargs):
            return self._resolve_forward_method()(*args, **kwargs)

    fo.BaseFusedOp = BaseFusedOp
    ln, act = created["sglang.srt.layers.layernorm"], created["sglang.srt.layers.activation"]
    rope, tk = created["sglang.srt.layers.rotary_embedding.base"], created["sglang.srt.layers.moe.topk"]

    class RMSNorm(BaseFusedOp):                                            # layernorm.py:423-480
        def __init__(self, hidden_size, eps=1e-6):
            super().__init__()
            self.weight = torch.nn.Parameter(torch.ones(hidden_size, dtype=torch.bfloat16), requires_grad=False)
            self.variance_epsilon, self.hidden_size = eps, hidden_size
            self.variance_size_override, self.cast_x_before_out_mul, self.fp32_residual = None, False, False

        def forward_native(self, x, residual=None, post_residual_addition=None, quant_linear=None):
            self.native_args = (residual is not None, post_residual_addition is not None, quant_linear is not None)
            return (x, residual) if residual is not None else x

    class TopK(BaseFusedOp):                                               # topk.py:392-520
        def __init__(self, cfg):
            super().__init__()
            self.topk_config = cfg

        def forward_native(self, hidden_states, router_logits, *, num_token_non_padded=None, expert_location_dispatch_info=None):
            return "native-topk"

    ln.RMSNorm, tk.TopK = RMSNorm, TopK
    act.SiluAndMul = type("SiluAndMul", (BaseFusedOp,), {"forward_native": lambda self, x: "native-silu"})
    rope.RotaryEmbedding = type("RotaryEmbedding", (BaseFusedOp,), {"forward_native": lambda self, *a, **k: "native-rope"})
    # rope_variant.py: the cache-only variants subclass RotaryEmbedding and define neither forward nor forward_native
    rv = created["sglang.srt.layers.rotary_embedding.rope_variant"]
    for vname in ("Llama3RotaryEmbedding", "DynamicNTKAlphaRotaryEmbedding", "DynamicNTKScalingRotaryEmbedding"):
        rec = ref("sglang.srt.layers.rotary_embedding.rope_variant", vname)
        assert rec["bases"] == ["RotaryEmbedding"] and "forward" not in rec["methods"] and "forward_native" not in rec["methods"], vname
        setattr(rv, vname, type(vname, (rope.RotaryEmbedding,), {}))
    # unquant.py:384: the MoE method is a BaseFusedOp whose forward_cuda asks the MoeRunner
    uq = created["sglang.srt.layers.quantization.unquant"]
    mrec = ref("sglang.srt.layers.quantization.unquant", "UnquantizedFusedMoEMethod")
    assert "BaseFusedOp" in mrec["bases"] and [p["name"] for p in mrec["methods"]["forward_cuda"]["params"]] == ["self", "layer", "dispatch_output"]
    uq.UnquantizedFusedMoEMethod = type("UnquantizedFusedMoEMethod", (BaseFusedOp,), {
        "forward_cuda": lambda self, layer, dispatch_output: ("runner.run", layer, dispatch_output),
        "forward_native": lambda self, layer, dispatch_output: "per-expert torch loop"})
    tk.StandardTopKOutput = __import__("collections").namedtuple("StandardTopKOutput", ref("sglang.srt.layers.moe.topk", "StandardTopKOutput")["__contract_fields__"]
                                                                   if False else [f["name"] for f in ref("sglang.srt.layers.moe.topk", "StandardTopKOutput")["fields"]])
    mp_ = created["sglang.srt.mem_cache.memory_pool"]
    mp_.KVWriteLoc = __import__("collections").namedtuple("KVWriteLoc", [f["name"] for f in ref("sglang.srt.mem_cache.memory_pool", "KVWriteLoc")["fields"]],
                                                          defaults=(None, None))
    created["sglang.srt.model_executor.runner.decode_cuda_graph_runner"].DecodeCudaGraphRunner = type("DecodeCudaGraphRunner", (), {})
    # allocator/__init__.py:3-8 re-exports the paged allocator; memory_pool's pool / helpers are the contract's classes
    module("sglang.srt.mem_cache.allocator").PagedTokenToKVPoolAllocator = created["sglang.srt.mem_cache.allocator.paged"].PagedTokenToKVPoolAllocator
    mp_.unwrap_write_loc = lambda loc_info: (tuple(loc_info) + (None, None))[:3] if isinstance(loc_info, tuple) else (loc_info, None, None)   # memory_pool.py:1586-1600
    plat_pkg._set = lambda p: current.__setitem__("platform", p)

    # ---- HookRegistry (srt/plugins/hook_registry.py:67-330), restated: register() records, apply_hooks() resolves the
    #      dotted target and wraps it; AROUND = hook(original_fn, *args, **kwargs) (:_wrap_fn) -------------------------
    hr = created["sglang.srt.plugins.hook_registry"]
    HookType = enum.Enum("HookType", {k: v for k, v in ref("sglang.srt.plugins.hook_registry", "HookType")["attrs"].items()})
    hr.HookType = HookType

    class HookRegistry:
        _hooks = {}
        _patched = set()

        @classmethod
        def register(cls, target, hook, hook_type=HookType.AFTER, *, source=None):
            if isinstance(hook, type) and hook_type != HookType.REPLACE:
                raise TypeError("class hooks need REPLACE")
            cls._hooks.setdefault(target, []).append((hook_type, hook, source))

        @classmethod
        def apply_hooks(cls):
            import functools
            import pkgutil

            for target, hooks in cls._hooks.items():
                if target in cls._patched:
                    continue
                obj_path, attr = target.rsplit(".", 1)
                obj = pkgutil.resolve_name(obj_path)
                wrapped = getattr(obj, attr)
                for ht, hook, _ in hooks:
                    assert ht == HookType.AROUND, "the stand-in restates AROUND only"

                    def wrapper(*args, __orig=wrapped, __hook=hook, **kwargs):
                        return __hook(__orig, *args, **kwargs)

                    wrapped = functools.wraps(wrapped)(wrapper)
                setattr(obj, attr, wrapped)
                cls._patched.add(target)

        @classmethod
        def reset(cls):
            cls._hooks.clear()
            cls._patched.clear()

    hr.HookRegistry = HookRegistry
    # forward_batch_info.py:1807-1816: the module-level name `clamp_position` is bound to one of two functions at import
    fbi_mod = created["sglang.srt.model_executor.forward_batch_info"]
    fbi_mod.clamp_position = fbi_mod._clamp_position_native
    created["sglang.srt.runtime_context"].get_parallel = lambda: types.SimpleNamespace(tp_size=1, tp_rank=0)
    return created


def test_plugin_load_runs_against_the_reference_contract(fake_sglang):
    from sglang_amd import platform, plugin
    from sglang_amd.layers.attention.hip_backend import HipAttnBackend

    plugin.load()
    g = fake_sglang
    # attention backend: registered under the CLI choice, factory takes a reference-shaped runner
    assert platform.BACKEND_NAME in g["sglang.srt.server_args"].ATTENTION_BACKEND_CHOICES
    factory = g["sglang.srt.layers.attention.attention_registry"].ATTENTION_BACKENDS[platform.BACKEND_NAME]

    class Pool:                                                           # memory_pool.py MHATokenToKVPool (fields read)
        start_layer = 0

        def __init__(self):
            self.k = torch.zeros((64, 2, 64), dtype=torch.bfloat16)

        def get_key_buffer(self, layer_id):
            return self.k

    class MC:                                                             # configs/model_config.py:1106-1196
        num_attention_heads, num_key_value_heads, context_len = 16, 4, 96

        def get_num_attention_heads(self, tensor_parallel_size):
            return max(1, self.num_attention_heads // tensor_parallel_size)

        def get_num_kv_heads(self, tensor_parallel_size, dcp_size=1):
            return max(1, self.num_key_value_heads // tensor_parallel_size)

    runner = types.SimpleNamespace(device=torch.device("cpu"), model_config=MC(), tp_size=2, sliding_window_size=None,
                                   token_to_kv_pool=Pool(), server_args=None,
                                   req_to_token_pool=types.SimpleNamespace(req_to_token=torch.zeros((9, 96), dtype=torch.int32), size=8))
    be = factory(runner)
    assert isinstance(be, HipAttnBackend) and (be.num_q_heads, be.num_kv_heads, be.head_dim, be.max_context_len) == (8, 2, 64, 96)
    # what the kernels cannot serve is refused when the server BUILDS the backend (model_runner.py init_attention_backend), by name
    for change, word in ((dict(dtype=torch.float16), "model dtype"), (dict(pool_dtype=torch.float16), "KV pool dtype"), (dict(head_dim=96), "head_dim 96")):
        bad_pool = Pool()
        if "head_dim" in change:
            bad_pool.k = torch.zeros((64, 2, 96), dtype=torch.bfloat16)
        if "pool_dtype" in change:
            bad_pool.dtype = change["pool_dtype"]
        bad = types.SimpleNamespace(**{**vars(runner), "token_to_kv_pool": bad_pool, **({"dtype": change["dtype"]} if "dtype" in change else {})})
        with pytest.raises(NotImplementedError, match=word):
            factory(bad)
    # ... and an instance of the REFERENCE's AttentionBackend: the runners read `shared_read_ends`, `supports_ragged_verify_graph`,
    # `on_after_cuda_graph_warmup` ... of a backend (decode_cuda_graph_runner.py:491, :724), which keep the reference's defaults
    ref_base = g["sglang.srt.layers.attention.base_attn_backend"].AttentionBackend
    assert isinstance(be, ref_base) and type(be).__mro__.index(HipAttnBackend) < type(be).__mro__.index(ref_base)
    for name in ("shared_read_ends", "supports_ragged_verify_graph", "on_after_cuda_graph_warmup", "supports_full_cuda_graph_chunked_prefix"):
        assert hasattr(be, name), name
    assert be.support_triton() is False and be.get_cuda_graph_seq_len_fill_value() == 1

    # sampler: a subclass of the reference Sampler with the gfx950 forward and the reference's own helpers
    smp = g["sglang.srt.layers.sampler"]
    s = smp._SAMPLER_FACTORIES[platform.BACKEND_NAME]()
    assert isinstance(s, smp.Sampler) and type(s).forward is not smp.Sampler.forward
    assert type(s)._sync_token_ids_across_tp is smp.Sampler._sync_token_ids_across_tp
    assert s.forward(types.SimpleNamespace(next_token_logits=torch.zeros((0, 8))), None, False, None, None, None).numel() == 0
    # what the reference computes differently on request stays its own forward, on the same instance (sampler.py:128,159-207,238)
    from sglang_amd import plugin as _plugin

    lo = types.SimpleNamespace(next_token_logits=torch.zeros((0, 8)))
    smp.SGLANG_RETURN_ORIGINAL_LOGPROB = False         # (the contract records the expression, `get_bool_env_var(...)`: unset = False)
    assert _plugin.sampler_declines(s, types.SimpleNamespace(return_sampling_masks=None), True) is None
    assert _plugin.sampler_declines(s, types.SimpleNamespace(return_sampling_masks=[False, False]), False) is None
    assert s.forward(lo, types.SimpleNamespace(return_sampling_masks=[False, True]), False, None, None, None) == "the reference's forward"
    s.rl_on_policy_target = "fsdp"
    assert s.forward(lo, types.SimpleNamespace(), False, None, None, None) == "the reference's forward"
    s.rl_on_policy_target = None
    s.enable_deterministic = True
    assert _plugin.sampler_declines(s, types.SimpleNamespace(), False) is None             # seeded sampling itself is the gfx950 kernel
    assert s.forward(lo, types.SimpleNamespace(), True, None, None, None) == "the reference's forward"
    s.enable_deterministic = False
    smp.SGLANG_RETURN_ORIGINAL_LOGPROB = True
    assert _plugin.sampler_declines(s, types.SimpleNamespace(), True) == "SGLANG_RETURN_ORIGINAL_LOGPROB"
    assert _plugin.sampler_declines(s, types.SimpleNamespace(), False) is None
    smp.SGLANG_RETURN_ORIGINAL_LOGPROB = False
    assert s.reference_forwards == 3 and s.forward(lo, None, False, None, None, None).numel() == 0

    # platform: out-of-tree, dispatch key = the key the forwards were registered under
    cls = platform._build_platform_class()
    iface = g["sglang.srt.platforms.interface"]
    assert issubclass(cls, iface.SRTPlatform)
    p = cls()
    assert p.is_out_of_tree() and p.get_dispatch_key_name() == platform.DISPATCH_KEY
    assert p.get_default_attention_backend() == platform.BACKEND_NAME and p.support_cuda_graph()
    assert p.get_graph_runner_cls().__name__ == "DecodeCudaGraphRunner"
    # the pool / allocator classes the reference instantiates are ITS classes with the gfx950 launches in place of the Triton / JIT ones
    alloc_cls, pool_cls = p.get_paged_allocator_cls(), p.get_mha_kv_pool_cls()
    assert issubclass(alloc_cls, g["sglang.srt.mem_cache.allocator"].PagedTokenToKVPoolAllocator) and alloc_cls.__name__.startswith("Mi355x")
    assert issubclass(pool_cls, g["sglang.srt.mem_cache.memory_pool"].MHATokenToKVPool) and pool_cls.__name__.startswith("Mi355x")
    import inspect as _inspect

    for cls_, mod_, base_, meths in ((alloc_cls, "sglang.srt.mem_cache.allocator.paged", "PagedTokenToKVPoolAllocator", ("alloc_extend", "alloc_decode")),
                                     (pool_cls, "sglang.srt.mem_cache.memory_pool", "MHATokenToKVPool", ("set_kv_buffer",))):
        for meth in meths:
            assert meth in vars(cls_), (cls_, meth)
            assert list(_inspect.signature(vars(cls_)[meth]).parameters)[1:] == [q_["name"] for q_ in ref_params(mod_, base_, meth)], (base_, meth)
    g["sglang.srt.platforms"]._set(p)

    # fused elementwise ops: BaseFusedOp dispatches to the registered forwards, which accept the reference call forms
    BaseFusedOp = g["sglang.kernels.fused_op"].BaseFusedOp
    reg = BaseFusedOp._oot_forward_registry[platform.DISPATCH_KEY]
    ln = g["sglang.srt.layers.layernorm"]
    assert set(c.__name__ for c in reg) == {"RMSNorm", "SiluAndMul", "RotaryEmbedding", "TopK", "Llama3RotaryEmbedding",
                                            "DynamicNTKAlphaRotaryEmbedding", "DynamicNTKScalingRotaryEmbedding", "UnquantizedFusedMoEMethod"}
    # the cache-only rope variants get RotaryEmbedding's registered forward (the registry is keyed by the exact class) ...
    rv = g["sglang.srt.layers.rotary_embedding.rope_variant"]
    assert reg[rv.Llama3RotaryEmbedding] is reg[g["sglang.srt.layers.rotary_embedding.base"].RotaryEmbedding]
    # ... and the MoE method dispatches to the reference's own forward_cuda (-> MoeRunner -> the fused function), not to its
    # forward_native (the per-expert torch loop an out-of-tree platform would otherwise get)
    uq = g["sglang.srt.layers.quantization.unquant"].UnquantizedFusedMoEMethod
    assert reg[uq] is uq.forward_cuda and uq()("layer", "dispatch")[0] == "runner.run"
    norm = ln.RMSNorm(32)
    x, res, pra = (torch.zeros((3, 32), dtype=torch.bfloat16) for _ in range(3))
    # communicator.py:722 call form: (hidden_states, residual, post_residual_addition); CPU tensors -> the op hands
    # the call to the instance's forward_native with the same four arguments
    out = norm(x, res, pra)
    assert isinstance(out, tuple) and norm.native_args == (True, True, False)
    norm(x, None, None, quant_linear=torch.nn.Identity())
    assert norm.native_args == (False, False, True)
    # TopK: reads topk_config (topk.py:215-232); grouped / biased routers stay with the reference
    tk = g["sglang.srt.layers.moe.topk"]
    cfg = types.SimpleNamespace(top_k=2, renormalize=True, use_grouped_topk=True, custom_routing_function=None,
                                correction_bias=None, scoring_func="softmax", num_fused_shared_experts=0,
                                apply_routed_scaling_factor_on_output=False, output_format=None)
    assert tk.TopK(cfg)(torch.zeros(2, 8), torch.zeros(2, 4)) == "native-topk"
    # SiluAndMul / RotaryEmbedding: what the kernels do not take (host tensors, other dtypes, int32 positions) is the bound
    # instance's own forward_native, not an exception
    act_m, rope_m = g["sglang.srt.layers.activation"], g["sglang.srt.layers.rotary_embedding.base"]
    assert act_m.SiluAndMul()(torch.zeros(2, 32, dtype=torch.float16)) == "native-silu"
    assert act_m.SiluAndMul()(torch.zeros(2, 32, dtype=torch.bfloat16)) == "native-silu"            # (host tensor)
    r = rope_m.RotaryEmbedding()
    r.cos_sin_cache = torch.zeros(8, 16)
    assert r(torch.zeros(2, dtype=torch.int64), torch.zeros(2, 32, dtype=torch.float16), torch.zeros(2, 16, dtype=torch.float16)) == "native-rope"
    # RMSNorm: a width / layout outside the kernel's 16-byte vectors is forward_native too
    odd = ln.RMSNorm(12)
    odd(torch.zeros(3, 12, dtype=torch.bfloat16))
    assert odd.native_args == (False, False, False)

    # fused MoE: the ("none", "triton") slot is ours, the reference's function is the fallback for what we do not cover
    pool = g["sglang.srt.layers.moe.moe_runner.base"].FusedOpPool
    fn = pool.get_fused_func("none", "triton")
    tri = g["sglang.srt.layers.moe.moe_runner.triton"]
    assert fn is not tri.fused_experts_none_to_triton
    fields = [f["name"] for f in ref("sglang.srt.layers.moe.moe_runner.triton", "TritonMoeQuantInfo")["fields"]]
    q = types.SimpleNamespace(**{f: None for f in fields})
    q.w13_weight, q.w2_weight = torch.zeros((2, 8, 4), dtype=torch.bfloat16), torch.zeros((2, 4, 4), dtype=torch.bfloat16)
    for f in fields:
        if f.startswith("use_") or f in ("per_channel_quant", "fuse_swiglu_interleaved"):
            setattr(q, f, False)
    q.use_fp8_w8a8 = True
    disp = types.SimpleNamespace(hidden_states=torch.zeros((2, 4), dtype=torch.bfloat16), topk_output=None, hidden_states_pre_quant=None)
    cfg_r = types.SimpleNamespace(activation="silu", is_gated=True, no_combine=False, apply_router_weight_on_input=False, inplace=False,
                                  routed_scaling_factor=None)
    assert fn(disp, q, cfg_r) == "reference-result" and tri.calls == ["reference"]
    # A MoeRunnerConfig carrying the REFERENCE'S OWN field defaults (base.py:37-65; among them gate_up_interleaved = True, which only
    # selects between the alpha / limit swiglu kernels) with unquantised bf16 weights is a call the gfx950 path takes ...
    defaults = {k: v for k, v in ref("sglang.srt.layers.moe.moe_runner.base", "MoeRunnerConfig")["attrs"].items()}
    assert defaults["gate_up_interleaved"] is True and defaults["activation"] == "silu" and defaults["is_gated"] is True
    cfg_default = types.SimpleNamespace(**defaults)
    q_plain = types.SimpleNamespace(**{f: (False if (f.startswith("use_") or f in ("per_channel_quant", "fuse_swiglu_interleaved")) else None) for f in fields})
    q_plain.w13_weight, q_plain.w2_weight = q.w13_weight, q.w2_weight
    assert plugin.outside_hip_moe(disp, q_plain, cfg_default) is None
    # ... and each of these is not
    for field, value, why in (("gemm1_alpha", 1.702, "gemm1_alpha"), ("swiglu_limit", 10.0, "swiglu_limit"), ("no_combine", True, "no_combine"),
                              ("activation", "gelu", "activation"), ("apply_router_weight_on_input", True, "apply_router_weight_on_input")):
        c2 = types.SimpleNamespace(**dict(defaults, **{field: value}))
        assert plugin.outside_hip_moe(disp, q_plain, c2) == why
    assert plugin.outside_hip_moe(disp, q, cfg_default) == "use_fp8_w8a8"
    q.use_fp8_w8a8, q.b13 = False, torch.zeros(2, 8)
    assert fn(disp, q, cfg_r) == "reference-result"                       # expert biases: not silently dropped
    # loading twice must not trip the pool's duplicate check (plugins are loaded once per process, but be safe)
    plugin.load()


def test_model_level_hook_is_registered_and_falls_through_to_the_reference_forward(fake_sglang):
    """plugin.load() registers the fused decode step as an AROUND hook on LlamaModel.forward through the reference's own
    HookRegistry (hook_registry.py:84 register / :146 apply_hooks).  After apply_hooks() the stand-in LlamaModel -- the
    reference's forward signature, the reference's attribute names -- is driven through the hook: whatever the hook
    does not own (here: CPU tensors, then prefill, pipeline proxies, captured layers, TP > 1) reaches the ORIGINAL
    forward with the original arguments.  The fused branch itself needs the GPU: tests/test_model_hook_gpu.py."""
    from sglang_amd import fused_decode, plugin

    g = fake_sglang
    plugin.load()
    hr = g["sglang.srt.plugins.hook_registry"]
    target = "sglang.srt.models.llama.LlamaModel.forward"
    assert fused_decode.HOOK_TARGETS == (target, "sglang.srt.models.qwen2.Qwen2Model.forward", "sglang.srt.models.mixtral.MixtralModel.forward")
    # the sparse-MoE form (round 5): MixtralModel.forward has LlamaModel.forward's signature, its layers carry the block as
    # `block_sparse_moe` with the pieces the fused layer calls -- gate, topk, experts (mixtral.py:57-118, 202-261)
    mx = "sglang.srt.models.mixtral"
    assert [(ht.name, h) for ht, h, _ in hr.HookRegistry._hooks[fused_decode.HOOK_TARGETS[2]]] == [("AROUND", fused_decode.llama_model_forward_hook)]
    import inspect as _insp

    assert list(_insp.signature(fused_decode.llama_model_forward_hook).parameters) == ["original"] + [q["name"] for q in ref(mx, "MixtralModel")["methods"]["forward"]["params"]]
    for cls, names in (("MixtralModel", ("embed_tokens", "layers", "norm", "pp_group")),
                       ("MixtralDecoderLayer", ("self_attn", "block_sparse_moe", "input_layernorm", "post_attention_layernorm")),
                       ("MixtralAttention", ("qkv_proj", "o_proj", "rotary_emb", "attn", "num_heads", "num_kv_heads", "head_dim")),
                       ("MixtralMoE", ("gate", "topk", "experts", "tp_size"))):
        have = set(ref(mx, cls)["instance_attrs"])
        assert set(names) <= have, (cls, sorted(set(names) - have))
    assert ("MixtralDecoderLayer", "MixtralAttention", "MixtralMoE") in fused_decode.MOE_FORMS
    # the hook binds to the reference's forward signature: (original, self, <reference parameters>) -- for every model
    # class it is registered on, whose layers carry the attribute names the fused loop reads
    import inspect

    for tgt, mod, prefix in ((target, "sglang.srt.models.llama", "Llama"), (fused_decode.HOOK_TARGETS[1], "sglang.srt.models.qwen2", "Qwen2")):
        assert [(ht.name, h) for ht, h, _ in hr.HookRegistry._hooks[tgt]] == [("AROUND", fused_decode.llama_model_forward_hook)]
        ref_names = [p["name"] for p in ref(mod, prefix + "Model")["methods"]["forward"]["params"]]
        assert list(inspect.signature(fused_decode.llama_model_forward_hook).parameters) == ["original"] + ref_names
        assert [p["name"] for p in ref(mod, prefix + "DecoderLayer")["methods"]["forward"]["params"]] == \
            ["self", "positions", "hidden_states", "forward_batch", "residual"]
        for cls, names in ((prefix + "Model", ("embed_tokens", "layers", "norm", "pp_group", "layers_to_capture")),
                           (prefix + "DecoderLayer", ("self_attn", "mlp", "input_layernorm", "post_attention_layernorm")),
                           (prefix + "Attention", ("qkv_proj", "o_proj", "rotary_emb", "attn", "num_heads", "num_kv_heads", "head_dim")),
                           (prefix + "MLP", ("gate_up_proj", "down_proj", "act_fn"))):
            have = set(ref(mod, cls)["instance_attrs"])
            assert set(names) <= have, (cls, sorted(set(names) - have))

    llama = g["sglang.srt.models.llama"]
    calls = []

    def reference_forward(self, input_ids, positions, forward_batch, input_embeds=None, pp_proxy_tensors=None):
        calls.append((input_ids, positions, forward_batch, input_embeds, pp_proxy_tensors))
        return "reference-forward"

    llama.LlamaModel.forward = reference_forward
    hr.HookRegistry.apply_hooks()
    assert llama.LlamaModel.forward is not reference_forward

    from sglang_amd.harness.models import CONFIGS, CausalLM

    inner = CausalLM(CONFIGS["tiny-llama"], torch.device("cpu"))          # the reference's attribute names on every level
    model = llama.LlamaModel.__new__(llama.LlamaModel)
    model.layers, model.norm, model.layers_to_capture = inner.layers, inner.norm, []
    model.embed_tokens = lambda ids: torch.nn.functional.embedding(ids, inner.embed_tokens)
    model.pp_group = types.SimpleNamespace(is_first_rank=True, is_last_rank=True)
    model.start_layer, model.end_layer = 0, len(inner.layers)
    decode = types.SimpleNamespace(forward_mode=types.SimpleNamespace(is_decode=lambda: True))
    extend = types.SimpleNamespace(forward_mode=types.SimpleNamespace(is_decode=lambda: False))
    ids, pos = torch.tensor([1, 2, 3]), torch.tensor([5, 6, 7])
    # decode batch on CPU tensors: the fused form does not apply -> the reference's own forward, same arguments
    assert model.forward(ids, pos, decode) == "reference-forward" and calls[-1][:3] == (ids, pos, decode)
    assert model.forward(ids, pos, extend) == "reference-forward"
    assert model.forward(ids, pos, decode, input_embeds=torch.zeros(3, 4)) == "reference-forward" and calls[-1][3] is not None
    assert model.forward(ids, pos, decode, None, {"hidden_states": None}) == "reference-forward"
    model.layers_to_capture = [1]
    assert model.forward(ids, pos, decode) == "reference-forward"
    model.layers_to_capture = []
    # TP > 1: the hook asks the reference's TP group for its xGMI communicator (tp_hooks.attach builds it inside
    # GroupCoordinator.__init__); a group without one -> the reference's own layer loop
    psm = g["sglang.srt.distributed.parallel_state"]
    psm.get_tp_group = lambda: types.SimpleNamespace(world_size=2)
    assert fused_decode._tp() == (2, None)
    assert model.forward(ids, pos, decode) == "reference-forward"
    psm.get_tp_group = lambda: types.SimpleNamespace(world_size=1)
    assert fused_decode._tp() == (1, None)
    assert fused_decode._reference_model_applies(model, decode, None, None)
    # the layer test the hook applies: dense unquantised neox layers qualify, a quantised projection does not
    layer = inner.layers[0]
    assert not fused_decode.layer_fusable(layer, 4)                        # CPU weights
    assert len(calls) == 6
    hr.HookRegistry.reset()


def test_tp_hooks_are_registered_on_the_group_coordinator_and_route_only_what_the_kernels_take(fake_sglang):
    """VERDICT r03 missing #3: TP > 1 through the drop-in surface.  plugin.load() registers AROUND hooks on
    GroupCoordinator.__init__ / .all_reduce / .fused_allreduce_rmsnorm / .all_gather (parallel_state.py:278, :648, :774,
    :1273); each hook binds to the reference method's own parameter list.  After apply_hooks() a stand-in group -- the
    reference's constructor signature and attribute names -- is driven through them: no communicator (CPU group, or the
    reference built the group without `use_custom_allreduce`) means the reference's method with the original arguments;
    with a communicator attached, bf16 messages inside its size range go to it, everything else still to the
    reference.  The real kernels behind a real communicator: tests/test_tp_hooks_gpu.py (two processes)."""
    import inspect

    from sglang_amd import plugin, tp_hooks

    g = fake_sglang
    plugin.load()
    hr = g["sglang.srt.plugins.hook_registry"]
    psm = g["sglang.srt.distributed.parallel_state"]
    GC = psm.GroupCoordinator
    gc_ref = ref("sglang.srt.distributed.parallel_state", "GroupCoordinator")
    hooks = dict(zip(tp_hooks.HOOK_TARGETS, tp_hooks._HOOKS))
    for target, hook in hooks.items():
        assert [(ht.name, h) for ht, h, _ in hr.HookRegistry._hooks[target]] == [("AROUND", hook)], target
        meth = target.rsplit(".", 1)[1]
        ref_ps = gc_ref["methods"][meth]["params"]
        ours = list(inspect.signature(hook).parameters.values())
        assert ours[0].name == "original" and ours[1].name == "self"
        if meth == "__init__":                       # forwarded verbatim: (original, self, *args, **kwargs)
            assert [p.kind for p in ours[2:]] == [inspect.Parameter.VAR_POSITIONAL, inspect.Parameter.VAR_KEYWORD]
        else:
            assert [p.name for p in ours[1:]] == [p["name"] for p in ref_ps], (meth, ref_ps)
            assert [p.default is not inspect.Parameter.empty for p in ours[1:]] == [p["default"] for p in ref_ps], meth
    # every attribute the hooks read is one the reference's constructor binds
    for a in ("world_size", "rank_in_group", "ranks", "cpu_group", "device_group", "device", "use_custom_allreduce", "unique_name"):
        assert a in gc_ref["instance_attrs"], a
    assert "get_tp_group" in MODS["sglang.srt.distributed.parallel_state"]["names"]
    for fn in ("tensor_model_parallel_all_reduce", "tensor_model_parallel_fused_allreduce_rmsnorm", "tensor_model_parallel_all_gather"):
        assert fn in MODS["sglang.srt.distributed.communication_op"]["names"], fn

    # ---- the reference's behaviour, restated where the hooks depend on it ---------------------------------------
    calls = []

    def ref_init(self, group_ranks, local_rank, torch_distributed_backend, use_pynccl, use_pymscclpp, use_custom_allreduce,
                 use_torch_symm_mem_all_reduce, use_hpu_communicator, use_xpu_communicator, use_npu_communicator,
                 use_message_queue_broadcaster=False, group_name=None, **kw):          # parallel_state.py:278-420
        self.unique_name = f"{group_name or 'anonymous'}:0"
        self.ranks, self.world_size, self.rank_in_group = group_ranks[0], len(group_ranks[0]), 0
        self.cpu_group, self.device_group, self.device = "gloo-group", "rccl-group", torch.device("cpu")
        self.use_custom_allreduce = use_custom_allreduce
        calls.append(("init", group_name))

    assert [p["name"] for p in gc_ref["methods"]["__init__"]["params"]][:11] == list(inspect.signature(ref_init).parameters)[:11]
    GC.__init__ = ref_init
    GC.all_reduce = lambda self, input_: calls.append(("all_reduce", input_)) or "reference-all-reduce"
    GC.fused_allreduce_rmsnorm = lambda self, input_, residual_inp_, weight_, eps: calls.append(("fused", input_)) or None
    GC.all_gather = lambda self, input_, dim=-1, output_tensor_list=None: calls.append(("all_gather", input_, dim)) or "reference-all-gather"
    hr.HookRegistry.apply_hooks()

    grp = GC([[0, 1]], 0, "nccl", False, False, True, False, False, False, False, group_name="tp")
    assert calls[0] == ("init", "tp")
    assert tp_hooks.communicator_of(grp) is None            # a CPU group: attach() declines without touching the groups
    x = torch.zeros((4, 64), dtype=torch.bfloat16)
    assert grp.all_reduce(x) == "reference-all-reduce" and calls[-1][0] == "all_reduce" and calls[-1][1] is x
    assert grp.fused_allreduce_rmsnorm(x, x.clone(), torch.ones(64, dtype=torch.bfloat16), 1e-5) is None and calls[-1][0] == "fused"
    assert grp.all_gather(x) == "reference-all-gather" and calls[-1][2] == -1

    class Comm:                                              # the XgmiAllReduce surface the hooks use
        disabled = False
        log = []

        def should_use(self, t):
            return t.dtype == torch.bfloat16 and t.numel() * 2 <= 2048

        def should_use_two_stage(self, t):
            return t.dtype == torch.bfloat16 and t.numel() * 2 <= 8192

        def all_reduce_any(self, t):
            self.log.append("all_reduce_any"); return "xgmi-sum"

        def all_reduce_add_rmsnorm(self, t, residual, w, eps):
            self.log.append(("add_rmsnorm", eps)); return "xgmi-normed"

        def fits_all_gather(self, t):
            return True

        def all_gather(self, t):
            self.log.append("all_gather"); return "xgmi-gathered"

    setattr(grp, tp_hooks.XGMI_ATTR, Comm())
    n = len(calls)
    assert grp.all_reduce(x) == "xgmi-sum"
    res = x.clone()
    assert grp.fused_allreduce_rmsnorm(x, res, torch.ones(64, dtype=torch.bfloat16), 1e-5) == ("xgmi-normed", res)
    assert len(calls) == n                                   # the reference's methods were not entered
    # outside the kernels' range -> the reference's own method, same arguments
    big = torch.zeros((64, 128), dtype=torch.bfloat16)      # 16 KiB: beyond this stand-in communicator's two-stage limit
    assert grp.all_reduce(big) == "reference-all-reduce" and calls[-1][1] is big
    f32 = torch.zeros((4, 64))
    assert grp.all_reduce(f32) == "reference-all-reduce"
    assert grp.fused_allreduce_rmsnorm(f32, f32.clone(), torch.ones(64), 1e-5) is None
    assert grp.all_gather(x, 0) == "reference-all-gather"   # (gathers along the rows stay the reference's; CPU tensors too)
    assert grp.all_gather(x) == "reference-all-gather"
    assert Comm.log == ["all_reduce_any", ("add_rmsnorm", 1e-5)]
    # a group the reference built WITHOUT its custom all-reduce hint, or one that is not on the decode path, gets none
    for kw, flag in ((dict(group_name="tp"), False), (dict(group_name="pp"), True), (dict(group_name="world"), True)):
        other = GC([[0, 1]], 0, "nccl", False, False, flag, False, False, False, False, **kw)
        assert tp_hooks.communicator_of(other) is None
    hr.HookRegistry.reset()


def test_linear_hook_is_registered_on_the_unquantized_method_and_falls_through(fake_sglang):
    """Models without a model-level hook (Mixtral's attention, any other dense architecture) get the weight-streaming GEMM
    for their decode-sized projections through an AROUND hook on UnquantizedLinearMethod.apply (unquant.py:243-293): bound
    to the reference's parameter list; CPU tensors, prefill-sized batches, other dtypes and subclassed weights reach the
    reference's own method with the original arguments.  The streamed branch: tests/test_model_hook_gpu.py."""
    import inspect

    from sglang_amd import linear_hook, plugin

    g = fake_sglang
    plugin.load()
    hr = g["sglang.srt.plugins.hook_registry"]
    assert [(ht.name, h) for ht, h, _ in hr.HookRegistry._hooks[linear_hook.HOOK_TARGET]] == [("AROUND", linear_hook.unquant_apply_hook)]
    ref_ps = ref("sglang.srt.layers.quantization.unquant", "UnquantizedLinearMethod")["methods"]["apply"]["params"]
    ours = list(inspect.signature(linear_hook.unquant_apply_hook).parameters.values())
    assert [p.name for p in ours] == ["original"] + [p["name"] for p in ref_ps]
    assert [p.default is not inspect.Parameter.empty for p in ours[1:]] == [p["default"] for p in ref_ps]
    # the lm_head: LogitsProcessor._compute_lm_head (logits_processor.py:706-769) carries an AROUND hook with the reference's
    # parameter list; only the plain-matmul branch is taken over (the head's method is one of the reference's unquantised ones)
    assert [(ht.name, h) for ht, h, _ in hr.HookRegistry._hooks[linear_hook.LM_HEAD_HOOK_TARGET]] == [("AROUND", linear_hook.compute_lm_head_hook)]
    lp_ps = ref("sglang.srt.layers.logits_processor", "LogitsProcessor")["methods"]["_compute_lm_head"]["params"]
    ours_lp = list(inspect.signature(linear_hook.compute_lm_head_hook).parameters.values())
    assert [p.name for p in ours_lp] == ["original"] + [p["name"] for p in lp_ps]
    assert [p.default is not inspect.Parameter.empty for p in ours_lp[1:]] == [p["default"] for p in lp_ps]
    unq = str(ref("sglang.srt.layers.logits_processor", "_UNQUANTIZED_LM_HEAD_METHODS")["value"])
    assert "UnquantizedEmbeddingMethod" in unq and "UnquantizedLinearMethod" in unq
    LP = g["sglang.srt.layers.logits_processor"].LogitsProcessor
    lp_calls = []
    LP._compute_lm_head = lambda self, hidden_states, lm_head, embedding_bias=None: lp_calls.append((hidden_states, lm_head)) or "reference-lm-head"
    ULM = g["sglang.srt.layers.quantization.unquant"].UnquantizedLinearMethod
    calls = []
    ULM.apply = lambda self, layer, x, bias=None: calls.append((layer, x, bias)) or "reference-linear"
    hr.HookRegistry.apply_hooks()
    head = types.SimpleNamespace(weight=torch.nn.Parameter(torch.zeros((256, 128), dtype=torch.bfloat16), requires_grad=False), quant_method=None)
    lp = LP.__new__(LP)
    assert lp._compute_lm_head(torch.zeros((4, 128), dtype=torch.bfloat16), head) == "reference-lm-head" and lp_calls[-1][1] is head   # CPU tensors
    m = ULM.__new__(ULM)
    layer = types.SimpleNamespace(weight=torch.nn.Parameter(torch.zeros((256, 128), dtype=torch.bfloat16), requires_grad=False))
    x = torch.zeros((4, 128), dtype=torch.bfloat16)
    assert m.apply(layer, x) == "reference-linear" and calls[-1][1] is x                  # CPU tensors
    assert m.apply(layer, x, torch.zeros(256, dtype=torch.bfloat16)) == "reference-linear" and calls[-1][2] is not None
    # the predicate itself, with the device check out of the way: only what the kernel's tiling takes
    import unittest.mock as um

    gpu = lambda t: um.patch.object(type(t), "is_cuda", property(lambda self: True))   # noqa: E731
    with gpu(x):
        w = layer.weight
        assert linear_hook.takes(x, w, None)                                          # 4 rows, N = 256, K = 128
        assert linear_hook.takes(torch.zeros((2, 3, 128), dtype=torch.bfloat16), w, None)   # leading dims flatten
        assert not linear_hook.takes(torch.zeros((4096, 128), dtype=torch.bfloat16), w, None)      # prefill-sized: the library GEMM
        assert not linear_hook.takes(x.float(), w, None) and not linear_hook.takes(x, w.float(), None)
        assert not linear_hook.takes(x, torch.zeros((250, 128), dtype=torch.bfloat16), None)       # N % 16
        assert not linear_hook.takes(torch.zeros((4, 96), dtype=torch.bfloat16), torch.zeros((256, 96), dtype=torch.bfloat16), None)   # K % 128
        assert not linear_hook.takes(x, w, torch.zeros(256))                            # fp32 bias
        assert not linear_hook.takes(x[:, ::1].t().t()[:, :], torch.zeros((256, 256), dtype=torch.bfloat16)[:, ::2], None)   # strided weight
    hr.HookRegistry.reset()


def test_position_hooks_are_registered_on_the_two_module_functions_and_fall_through(fake_sglang):
    """forward_batch_info.clamp_position / compute_position (:871-896, :1771-1816) carry AROUND hooks bound to the reference's
    parameter lists; CPU tensors, float lengths and mixed dtypes reach the reference's own functions with the original
    arguments.  The kernel branch: tests/test_reference_model_gpu.py (under the reference's own ForwardBatch.init_new)."""
    import inspect

    from sglang_amd import plugin, position_hooks

    g = fake_sglang
    plugin.load()
    hr = g["sglang.srt.plugins.hook_registry"]
    for target, hook in zip(position_hooks.HOOK_TARGETS, position_hooks._HOOKS):
        assert [(ht.name, h) for ht, h, _ in hr.HookRegistry._hooks[target]] == [("AROUND", hook)]
    ref_ps = [p["name"] for p in ref("sglang.srt.model_executor.forward_batch_info", "compute_position")["params"]]
    assert list(inspect.signature(position_hooks.compute_position_hook).parameters) == ["original"] + ref_ps
    ref_ps = [p["name"] for p in ref("sglang.srt.model_executor.forward_batch_info", "_clamp_position_native")["params"]]
    assert list(inspect.signature(position_hooks.clamp_position_hook).parameters) == ["original"] + ref_ps
    fbi = g["sglang.srt.model_executor.forward_batch_info"]
    calls = []
    fbi.clamp_position = lambda seq_lens: calls.append(("clamp", seq_lens)) or "reference-clamp"
    fbi.compute_position = lambda attn_backend, extend_prefix_lens, extend_seq_lens, extend_seq_lens_sum: calls.append(
        ("compute", attn_backend, extend_prefix_lens, extend_seq_lens, extend_seq_lens_sum)) or "reference-compute"
    hr.HookRegistry.apply_hooks()
    lens = torch.tensor([5, 9], dtype=torch.int32)
    assert fbi.clamp_position(lens) == "reference-clamp" and calls[-1][1] is lens                     # CPU tensor
    assert fbi.compute_position("triton", lens, lens, 14) == "reference-compute" and calls[-1][1:] == ("triton", lens, lens, 14)
    import unittest.mock as um

    with um.patch.object(torch.Tensor, "is_cuda", property(lambda self: True)):
        assert not position_hooks._lens_ok(lens.float()) and not position_hooks._lens_ok(lens.view(1, 2)) and position_hooks._lens_ok(lens)
        assert fbi.compute_position("triton", lens, lens.long(), 14) == "reference-compute"            # mixed dtypes
        assert fbi.compute_position("triton", lens, lens, 0) == "reference-compute"                    # an empty extend
    hr.HookRegistry.reset()


def test_mem_hooks_are_registered_on_the_allocation_helpers_and_fall_through(fake_sglang):
    """allocation.write_cache_indices / get_last_loc (mem_cache/allocation.py:54-148) carry AROUND hooks bound to the reference's
    parameter lists; CPU tensors (the reference's torch-native runs) reach the reference's own functions with the original arguments.
    `support_triton` keys on the backend NAME in this reference (utils/common.py:1307-1308) -- which is why the hooks exist.
    The kernel branch: tests/test_reference_model_gpu.py (zero Triton launches in a prefill under the reference's scheduler)."""
    import inspect

    from sglang_amd import mem_hooks, plugin

    g = fake_sglang
    assert [q["name"] for q in ref("sglang.srt.utils.common", "support_triton")["params"]] == ["backend"]
    plugin.load()
    hr = g["sglang.srt.plugins.hook_registry"]
    for target, hook in zip(mem_hooks.HOOK_TARGETS, mem_hooks._HOOKS):
        assert [(ht.name, h) for ht, h, _ in hr.HookRegistry._hooks[target]] == [("AROUND", hook)]
        fn = target.rsplit(".", 1)[1]
        assert list(inspect.signature(hook).parameters) == ["original"] + [q["name"] for q in ref("sglang.srt.mem_cache.allocation", fn)["params"]]
    al = g["sglang.srt.mem_cache.allocation"]
    calls = []
    al.write_cache_indices = lambda *a: calls.append(("write", a)) or "reference-write"
    al.get_last_loc = lambda *a: calls.append(("last", a)) or "reference-last"
    hr.HookRegistry.apply_hooks()
    i64 = torch.tensor([1, 2], dtype=torch.int64)
    pool = types.SimpleNamespace(req_to_token=torch.zeros((4, 8), dtype=torch.int32))
    args = (i64, i64, i64, i64, i64, i64, i64, i64, i64, [i64, i64], pool)
    assert al.write_cache_indices(*args) == "reference-write" and calls[-1] == ("write", args)            # CPU tensors
    assert al.get_last_loc(pool.req_to_token, i64, i64) == "reference-last" and calls[-1][0] == "last"
    import unittest.mock as um

    with um.patch.object(torch.Tensor, "is_cuda", property(lambda self: True)):
        assert mem_hooks._i64_cuda(i64) and not mem_hooks._i64_cuda(i64.int()) and not mem_hooks._i64_cuda(i64.view(1, 2))
        assert al.get_last_loc(pool.req_to_token.long(), i64, i64) == "reference-last"                    # an int64 table is not this path's
        assert al.write_cache_indices(*(args[:9] + ([i64.int(), i64], pool))) == "reference-write"        # int32 prefix slots
    hr.HookRegistry.reset()


def test_runner_config_and_quant_info_fields_the_moe_hook_reads_exist():
    cfg_fields = [f["name"] for f in ref("sglang.srt.layers.moe.moe_runner.base", "MoeRunnerConfig")["fields"]]
    for f in ("activation", "is_gated", "inplace", "no_combine", "routed_scaling_factor", "apply_router_weight_on_input"):
        assert f in cfg_fields, f
    q_fields = [f["name"] for f in ref("sglang.srt.layers.moe.moe_runner.triton", "TritonMoeQuantInfo")["fields"]]
    for f in ("w13_weight", "w2_weight", "b13", "b2", "use_fp8_w8a8", "fuse_swiglu_interleaved", "block_shape"):
        assert f in q_fields, f
    tk_fields = [f["name"] for f in ref("sglang.srt.layers.moe.topk", "TopKConfig")["fields"]]
    for f in ("top_k", "renormalize", "use_grouped_topk", "custom_routing_function", "correction_bias", "scoring_func"):
        assert f in tk_fields, f
    assert [f["name"] for f in ref("sglang.srt.layers.moe.topk", "StandardTopKOutput")["fields"]] == ["topk_weights", "topk_ids", "router_logits"]
    assert [f["name"] for f in ref("sglang.srt.mem_cache.memory_pool", "KVWriteLoc")["fields"]][:2] == ["loc", "swa_loc"]
    assert [p["name"] for p in ref_params("sglang.srt.mem_cache.memory_pool", "MHATokenToKVPool", "set_kv_buffer")][:4] == ["layer", "loc_info", "cache_k", "cache_v"]


def test_names_the_sampler_and_router_sides_read_exist(monkeypatch):
    """plugin.sampler_declines and layers/moe/topk._ReferenceSide read instance attributes, dataclass fields, a module constant and
    names the reference's topk module binds by IMPORT: all recorded by gen_contract.py, so a reference that renames one fails here
    after a regeneration instead of silently never declining / never reporting."""
    smp = ref("sglang.srt.layers.sampler", "Sampler")
    for a in ("rl_on_policy_target", "enable_deterministic", "use_log_softmax_logprob", "use_ascend_backend", "output_logprob_processor"):
        assert a in smp["instance_attrs"], a
    assert MODS["sglang.srt.layers.sampler"]["names"]["SGLANG_RETURN_ORIGINAL_LOGPROB"]["kind"] == "constant"
    sbi = [f["name"] for f in ref("sglang.srt.sampling.sampling_batch_info", "SamplingBatchInfo")["fields"]]
    for f in ("return_sampling_masks", "is_all_greedy", "need_top_p_sampling", "need_top_k_sampling", "need_min_p_sampling", "sampling_seed",
              "temperatures", "top_ps", "top_ks", "min_ps"):
        assert f in sbi, f
    tk = MODS["sglang.srt.layers.moe.topk"]["names"]
    for a in ("enable_waterfill", "waterfill_balancer", "layer_id", "topk_config"):
        assert a in tk["TopK"]["instance_attrs"], a
    assert [q["name"] for q in tk["capture_routed_experts_if_allowed"]["params"]] == ["topk_config", "layer_id", "topk_ids"]
    assert tk["import:get_global_expert_distribution_recorder"]["from"].endswith("expert_distribution.get_global_expert_distribution_recorder")
    assert tk["import:get_moe_runner_backend"]["kind"] == "import" and tk["import:envs"]["from"] == "sglang.srt.environ.envs"
    assert [q["name"] for q in MODS["sglang.srt.eplb.expert_distribution"]["names"]["ExpertDistributionRecorder.on_select_experts"]["params"]] == ["self", "topk_ids"]
    backend = ref("sglang.srt.layers.moe.utils", "MoeRunnerBackend")
    assert "is_auto" in backend["methods"] and "is_triton" in backend["methods"]
    tkc = [f["name"] for f in tk["TopKConfig"]["fields"]]
    for f in ("output_format", "num_fused_shared_experts", "apply_routed_scaling_factor_on_output", "allow_routed_experts_capture"):
        assert f in tkc, f

    # behaviour of the side object on a module shaped like the reference's
    from types import SimpleNamespace as NS

    from sglang_amd.layers.moe import topk as hip_topk

    log = []
    flag = lambda v: NS(get=lambda: v)                                                            # noqa: E731
    fake = types.ModuleType("sglang.srt.layers.moe.topk")
    fake.capture_routed_experts_if_allowed = lambda topk_config, layer_id, topk_ids: log.append(("capture", layer_id))
    fake.get_global_expert_distribution_recorder = lambda: NS(on_select_experts=lambda topk_ids: log.append(("record", tuple(topk_ids.shape))))
    state = dict(backend=NS(is_auto=lambda: True, is_triton=lambda: False))
    fake.get_moe_runner_backend = lambda: state["backend"]
    fake.envs = NS(SGLANG_SIMULATE_UNIFORM_EXPERTS=flag(False), SGLANG_SIMULATE_ROUND_ROBIN_EXPERTS=flag(False))
    for name in ("sglang", "sglang.srt", "sglang.srt.layers", "sglang.srt.layers.moe"):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    monkeypatch.setitem(sys.modules, "sglang.srt.layers.moe.topk", fake)
    sys.modules["sglang.srt.layers.moe"].topk = fake
    sys.modules["sglang.srt.layers"].moe = sys.modules["sglang.srt.layers.moe"]
    sys.modules["sglang.srt"].layers = sys.modules["sglang.srt.layers"]
    sys.modules["sglang"].srt = sys.modules["sglang.srt"]
    side = hip_topk._ReferenceSide()
    cfg = NS(output_format=None)
    assert side.standard_output_expected(cfg)
    state["backend"] = NS(is_auto=lambda: False, is_triton=lambda: False)                        # e.g. --moe-runner-backend triton_kernel
    assert not side.standard_output_expected(cfg)
    assert side.standard_output_expected(NS(output_format="STANDARD"))                            # a config that names the format decides
    fake.envs.SGLANG_SIMULATE_ROUND_ROBIN_EXPERTS = flag(True)
    assert not side.standard_output_expected(NS(output_format="STANDARD"))
    side.after_select(cfg, 3, torch.zeros((5, 2), dtype=torch.int32))
    assert log == [("capture", 3), ("record", (5, 2))]
    # a forward bound to this package's own TopK has no reference side
    assert hip_topk._reference_side(hip_topk.TopK(2)) is None
