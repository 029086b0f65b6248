#!/usr/bin/env python3
"""Build the native (HIP/C++) parts of kvmini in-tree for gfx950.

* ``kvmini/_C.<abi>.so``       torch extension: HIP kernels (csrc/*.hip) + bindings.cpp
* ``kvmini/libkvmini_rt.so``      torch-free amd-smi sampler thread, loaded through ctypes
* ``kvmini/libkvmini_gather.so``  RCCL telemetry all-gather (binds to the process's HIP/RCCL at load)

hipcc cross-compiles for gfx950 without a GPU, so this runs on a CPU-only build
host; objects are cached under ``build/`` by source mtime.  ``python build_native.py``
builds everything, ``--force`` rebuilds, ``--only ext|rt`` picks one target.
"""

from __future__ import annotations

import argparse
import os
import shlex
import subprocess
import sys
import sysconfig
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
BUILD = ROOT / "build"
ARCH = os.environ.get("KVMINI_GPU_ARCH", "gfx950")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", "g++")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

KERNEL_SOURCES = ["hist_quantile.hip", "energy_trapz.hip", "engine_kernels.hip", "stream_gemm.hip", "wide_gemm.hip", "mfma_scaled.hip", "sampling.hip", "guided.hip", "xgmi_collectives.hip"]


def run(cmd: list[str]) -> None:
    print("+", " ".join(shlex.quote(c) for c in cmd), flush=True)
    subprocess.run(cmd, check=True)


def stale(out: Path, deps: list[Path]) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def ext_path() -> Path:
    return ROOT / "kvmini" / ("_C" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_ext(force: bool = False) -> Path:
    import torch
    from torch.utils import cpp_extension as ce

    BUILD.mkdir(exist_ok=True)
    objs = []
    for src in KERNEL_SOURCES:
        s = CSRC / src
        o = BUILD / (s.stem + ".o")
        if force or stale(o, [s]):
            run([HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", str(s), "-o", str(o)])
        objs.append(o)
    # bindings: torch headers (slow to parse) — compiled as HIP so the HIP runtime types resolve
    s = CSRC / "bindings.cpp"
    o = BUILD / "bindings.o"
    if force or stale(o, [s]):
        inc = []
        for p in ce.include_paths():
            inc += ["-isystem", p]
        inc += ["-isystem", sysconfig.get_paths()["include"]]
        run(
            [HIPCC, f"--offload-arch={ARCH}", "-x", "hip", "-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2",
             f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-DTORCH_EXTENSION_NAME=_C", "-DTORCH_API_INCLUDE_EXTENSION_H",
             "-Wno-deprecated-declarations", *inc, "-c", str(s), "-o", str(o)]
        )
    objs.append(o)
    out = ext_path()
    if force or stale(out, objs):
        libdirs = ce.library_paths()
        link = []
        for d in libdirs:
            link += [f"-L{d}", f"-Wl,-rpath,{d}"]
        # linked with the host C++ driver, not hipcc: hipcc would add the SYSTEM libamdhip64 next
        # to the one bundled with torch, and a process with two HIP runtimes fails in odd ways
        # (kernel launches of the second one report "no ROCm-capable device")
        run([CXX, "-shared", "-fPIC", *map(str, objs), *link, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python", "-lamdhip64", "-o", str(out)])
    return out


def rt_path() -> Path:
    return ROOT / "kvmini" / "libkvmini_rt.so"


def gather_path() -> Path:
    return ROOT / "kvmini" / "libkvmini_gather.so"


def build_rt(force: bool = False) -> Path:
    """Two torch-free libraries.  ``libkvmini_rt.so`` (amd-smi sampler thread, which opens
    libamd_smi with dlopen, and the analysis host routines) links no GPU runtime at all, so it can be loaded into any process
    at any time.  ``libkvmini_gather.so`` (RCCL telemetry all-gather) uses HIP + RCCL but is
    linked WITHOUT them: its loader (kvmini/telemetry/native_rt.py) first makes the process's
    one HIP runtime and RCCL global — the copies bundled with torch when torch is imported,
    the system ones otherwise — so a process never ends up with two HIP runtimes."""
    BUILD.mkdir(exist_ok=True)
    out = rt_path()
    srcs = [s for s in (CSRC / "amdsmi_sampler.cpp", CSRC / "host_routines.cpp") if s.exists()]
    if srcs and (force or stale(out, srcs)):
        run([CXX, "-O2", "-std=c++17", "-fPIC", "-shared", *map(str, srcs), "-lpthread", "-ldl", "-o", str(out)])
    gout = gather_path()
    gsrc = CSRC / "telemetry_gather.cpp"
    if gsrc.exists() and (force or stale(gout, [gsrc])):
        run([CXX, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", f"-I{ROCM}/include", str(gsrc), "-lpthread", "-o", str(gout)])
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--only", choices=["ext", "rt"], default=None)
    a = ap.parse_args()
    if a.only in (None, "ext"):
        print("built", build_ext(a.force))
    if a.only in (None, "rt"):
        print("built", build_rt(a.force))
    return 0


if __name__ == "__main__":
    sys.exit(main())
