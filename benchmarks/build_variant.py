"""Build an experiment variant of the library: `python benchmarks/build_variant.py <name> -DFLAG ...` compiles every
csrc/ translation unit with the extra flags into benchmarks/variants/lib_<name>.so (git-ignored as a build artefact,
not gpurun-ignored: it travels to the GPU box).  Experiments select it with SGLANG_AMD_LIB."""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sglang_amd import build as B  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    out = ROOT / "benchmarks" / "variants"
    obj = out / f"obj_{name}"
    obj.mkdir(parents=True, exist_ok=True)
    incs = [f"-I{d}" for d in (B.INCLUDE, B.PKG.parent / "include") if (d / "sglang_amd.h").exists()]

    def one(src):
        o = obj / (src.name + ".o")
        cmd = [B._hipcc(), *B.HIPCC_FLAGS, *flags, *incs, "-c", str(src), "-o", str(o)]
        if src.suffix == ".cpp":
            cmd[1:1] = ["-x", "hip"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(f"{src.name}:\n{r.stderr}")
        return o

    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(one, B.sources()))
    lib = out / f"lib_{name}.so"
    r = subprocess.run([B._hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", *map(str, objs), "-o", str(lib)], capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stderr)
    print("built", lib)


if __name__ == "__main__":
    main()
