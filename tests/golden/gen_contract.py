"""Extract the drop-in boundary's contract from the REAL reference sources (build container only).

    python tests/golden/gen_contract.py          # needs /root/reference; writes tests/golden/reference_contract.json

sglang cannot be imported here (orjson / msgspec / zmq missing), so the files SURVEY.md section 8(b) cites are
ast-parsed instead: for every module the plugin touches, the signatures (parameter names, kinds, which have
defaults) of the functions / methods it calls or must be call-compatible with, the fields of the dataclasses it reads,
class bases and class-level constants.  tests/test_plugin_contract.py builds a stand-in `sglang` package from this
file and runs plugin.load() against it; nothing under tests/ reads /root/reference at test time.
"""
from __future__ import annotations

import ast
import json
from pathlib import Path

REF = Path("/root/reference/python")
OUT = Path(__file__).resolve().parent / "reference_contract.json"

# module -> names to record ("Class.method" for methods, "Class" for the class record, "NAME" for constants)
WANT = {
    "sglang/srt/plugins/__init__.py": ["GENERAL_PLUGINS_GROUP", "PLATFORM_PLUGINS_GROUP", "load_plugins", "load_plugins_by_group"],
    "sglang/srt/platforms/interface.py": ["SRTPlatform"],
    "sglang/srt/platforms/device_mixin.py": ["DeviceMixin", "PlatformEnum"],
    "sglang/srt/platforms/cuda.py": ["CudaDeviceMixin"],
    "sglang/srt/platforms/__init__.py": ["_resolve_platform", "_load_platform_class"],
    "sglang/srt/layers/attention/attention_registry.py": ["register_attention_backend", "ATTENTION_BACKENDS"],
    "sglang/srt/server_args.py": ["add_attention_backend_choices", "ATTENTION_BACKEND_CHOICES"],
    "sglang/srt/layers/attention/base_attn_backend.py": ["AttentionBackend"],
    "sglang/srt/layers/attention/torch_native_backend.py": ["TorchNativeAttnBackend"],
    "sglang/srt/layers/radix_attention.py": ["RadixAttention", "AttentionType"],
    "sglang/srt/layers/sampler.py": ["Sampler", "register_sampler_backend", "create_sampler", "SGLANG_RETURN_ORIGINAL_LOGPROB"],
    "sglang/srt/sampling/sampling_batch_info.py": ["SamplingBatchInfo"],
    "sglang/srt/layers/moe/moe_runner/base.py": ["FusedOpPool", "register_fused_func", "MoeRunnerConfig"],
    "sglang/srt/layers/moe/moe_runner/triton.py": ["fused_experts_none_to_triton", "TritonMoeQuantInfo"],
    "sglang/srt/layers/moe/token_dispatcher/standard.py": ["StandardCombineInput", "StandardDispatchOutput"],
    "sglang/srt/layers/moe/topk.py": ["TopK", "TopKConfig", "StandardTopKOutput", "select_experts", "capture_routed_experts_if_allowed",
                                      "import:get_global_expert_distribution_recorder", "import:get_moe_runner_backend", "import:envs"],
    "sglang/srt/layers/moe/utils.py": ["MoeRunnerBackend"],
    "sglang/srt/eplb/expert_distribution.py": ["ExpertDistributionRecorder.on_select_experts"],
    "sglang/kernels/fused_op.py": ["BaseFusedOp", "_oot_dispatch_key"],
    "sglang/srt/layers/layernorm.py": ["RMSNorm"],
    "sglang/srt/layers/activation.py": ["SiluAndMul"],
    "sglang/srt/layers/rotary_embedding/base.py": ["RotaryEmbedding"],
    "sglang/srt/layers/rotary_embedding/rope_variant.py": ["Llama3RotaryEmbedding", "DynamicNTKAlphaRotaryEmbedding", "DynamicNTKScalingRotaryEmbedding"],
    "sglang/kernels/ops/attention/rope.py": ["FusedSetKVBufferArg"],
    "sglang/srt/mem_cache/memory_pool.py": ["KVWriteLoc", "MHATokenToKVPool", "ReqToTokenPool", "unwrap_write_loc", "_set_kv_buffer_impl"],
    "sglang/srt/mem_cache/allocation.py": ["write_cache_indices", "get_last_loc", "alloc_for_extend", "alloc_for_decode"],
    "sglang/srt/mem_cache/allocator/paged.py": ["PagedTokenToKVPoolAllocator"],
    "sglang/srt/utils/common.py": ["get_num_new_pages", "support_triton"],
    "sglang/srt/layers/logits_processor.py": ["LogitsProcessor", "should_apply_lm_head_quant_method", "_UNQUANTIZED_LM_HEAD_METHODS"],
    "sglang/srt/mem_cache/radix_cache.py": ["RadixCache", "RadixKey"],
    "sglang/srt/model_executor/forward_batch_info.py": ["ForwardBatch", "ForwardMode", "compute_position", "_clamp_position_native"],
    "sglang/srt/configs/model_config.py": ["ModelConfig.get_num_attention_heads", "ModelConfig.get_num_kv_heads"],
    "sglang/srt/model_executor/runner/decode_cuda_graph_runner.py": ["DecodeCudaGraphRunner"],
    "sglang/srt/plugins/hook_registry.py": ["HookRegistry", "HookType", "_wrap_fn"],
    "sglang/srt/models/llama.py": ["LlamaModel", "LlamaDecoderLayer", "LlamaAttention", "LlamaMLP"],
    "sglang/srt/models/qwen2.py": ["Qwen2Model", "Qwen2DecoderLayer", "Qwen2Attention", "Qwen2MLP"],
    "sglang/srt/models/mixtral.py": ["MixtralModel", "MixtralDecoderLayer", "MixtralAttention", "MixtralMoE"],
    "sglang/srt/layers/quantization/unquant.py": ["UnquantizedLinearMethod", "UnquantizedFusedMoEMethod"],
    "sglang/srt/runtime_context.py": ["get_parallel"],
    "sglang/srt/distributed/parallel_state.py": ["GroupCoordinator", "get_tp_group"],
    "sglang/srt/distributed/communication_op.py": ["tensor_model_parallel_all_reduce", "tensor_model_parallel_fused_allreduce_rmsnorm",
                                                   "tensor_model_parallel_all_gather"],
}
# sgl_kernel functional namespace (kernels/aot/python/sgl_kernel)
KERNEL_NS = {
    "python/sglang/kernels/aot/python/sgl_kernel/elementwise.py": ["rmsnorm", "fused_add_rmsnorm", "silu_and_mul", "rotary_embedding"],
    "python/sglang/kernels/aot/python/sgl_kernel/moe.py": ["topk_softmax", "moe_align_block_size", "moe_sum_reduce"],
    "python/sglang/kernels/aot/python/sgl_kernel/sampling.py": ["top_k_renorm_prob", "top_p_renorm_prob", "top_k_renorm_probs", "top_p_renorm_probs"],
}


def sig(fn: ast.FunctionDef):
    a = fn.args
    params = []
    pos = a.posonlyargs + a.args
    n_def = len(a.defaults)
    for i, p in enumerate(pos):
        params.append({"name": p.arg, "kind": "pos", "default": i >= len(pos) - n_def})
    if a.vararg:
        params.append({"name": a.vararg.arg, "kind": "var", "default": True})
    for p, d in zip(a.kwonlyargs, a.kw_defaults):
        params.append({"name": p.arg, "kind": "kw", "default": d is not None})
    if a.kwarg:
        params.append({"name": a.kwarg.arg, "kind": "varkw", "default": True})
    decos = [ast.unparse(d) for d in fn.decorator_list]
    return {"params": params, "decorators": decos, "line": fn.lineno}


def const(node):
    try:
        v = ast.literal_eval(node)
        json.dumps(v)
        return v
    except Exception:
        return ast.unparse(node)[:200]


def class_record(c: ast.ClassDef):
    rec = {"bases": [ast.unparse(b) for b in c.bases], "line": c.lineno, "methods": {}, "attrs": {}, "fields": []}
    for n in c.body:
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef)):
            rec["methods"][n.name] = sig(n)
        elif isinstance(n, ast.AnnAssign) and isinstance(n.target, ast.Name):
            rec["fields"].append({"name": n.target.id, "default": n.value is not None})
            if n.value is not None:
                rec["attrs"][n.target.id] = const(n.value)
        elif isinstance(n, ast.Assign):
            for t in n.targets:
                if isinstance(t, ast.Name):
                    rec["attrs"][t.id] = const(n.value)
    # names the constructor binds on the instance (`self.x = ...` anywhere in __init__): what code written against the
    # class reads as attributes
    inst = set()
    for n in c.body:
        if isinstance(n, ast.FunctionDef) and n.name == "__init__":
            for a in ast.walk(n):
                if isinstance(a, (ast.Assign, ast.AnnAssign, ast.AugAssign)):
                    for t in (a.targets if isinstance(a, ast.Assign) else [a.target]):
                        for e in (t.elts if isinstance(t, ast.Tuple) else [t]):
                            if isinstance(e, ast.Attribute) and isinstance(e.value, ast.Name) and e.value.id == "self":
                                inst.add(e.attr)
    rec["instance_attrs"] = sorted(inst)
    return rec


def extract(path: Path, names):
    tree = ast.parse(path.read_text())
    top = {}
    for n in tree.body:
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            top[n.name] = n
        elif isinstance(n, ast.Assign):
            for t in n.targets:
                if isinstance(t, ast.Name):
                    top[t.id] = n
        elif isinstance(n, ast.AnnAssign) and isinstance(n.target, ast.Name):
            top[n.target.id] = n
    imported = {}
    for n in tree.body:                      # names a module binds by importing them (read as `module.name` by code written against it)
        if isinstance(n, ast.ImportFrom):
            for al in n.names:
                imported[al.asname or al.name] = f"{'.' * n.level}{n.module or ''}.{al.name}"
    out = {}
    for name in names:
        if name.startswith("import:"):
            if name[7:] not in imported:
                raise SystemExit(f"{path}: no longer imports {name[7:]}")
            out[name] = {"kind": "import", "from": imported[name[7:]]}
            continue
        cls, _, meth = name.partition(".")
        node = top.get(cls)
        if node is None:
            raise SystemExit(f"{path}: {cls} not found -- the reference moved; update SURVEY section 8(b) citations")
        if isinstance(node, ast.ClassDef):
            rec = class_record(node)
            if meth:
                out[name] = {"kind": "method", **rec["methods"][meth]}
            else:
                out[name] = {"kind": "class", **rec}
        elif isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
            out[name] = {"kind": "function", **sig(node)}
        else:
            out[name] = {"kind": "constant", "value": const(node.value), "line": node.lineno}
    return out


def main():
    contract = {"source": "ast of /root/reference/python (sglang) and /root/reference/kernels/aot/python (sgl_kernel)",
                "modules": {}, "sgl_kernel": {}}
    for rel, names in WANT.items():
        contract["modules"][rel[:-3].replace("/__init__", "").replace("/", ".")] = {
            "file": rel, "is_package": rel.endswith("__init__.py"), "names": extract(REF / rel, names)}
    for rel, names in KERNEL_NS.items():
        contract["sgl_kernel"][rel] = extract(Path("/root/reference") / rel, names)
    OUT.write_text(json.dumps(contract, indent=1, sort_keys=True))
    n = sum(len(m["names"]) for m in contract["modules"].values())
    print(f"wrote {OUT} ({n} names from {len(contract['modules'])} modules)")


if __name__ == "__main__":
    main()
