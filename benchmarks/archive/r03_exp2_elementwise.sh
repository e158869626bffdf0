#!/bin/bash
# elementwise kernels at prefill sizes: default build vs variants (benchmarks/build_variant.py el_*)
for v in default el_oldnorm el_nt el_vpt8 el_nt8 el_vpt2; do
  echo "== $v"
  if [ $v = default ]; then timeout 120 python benchmarks/r02_exp12_elementwise.py 2>&1 | grep -v amdgpu.ids
  else SGLANG_AMD_LIB=benchmarks/variants/lib_$v.so timeout 120 python benchmarks/r02_exp12_elementwise.py 2>&1 | grep -v amdgpu.ids; fi
done
