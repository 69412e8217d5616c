"""Build libsome_amd.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

No torch / pybind involvement: the library is a plain C-ABI shared object (include/some_amd.h) loaded with
ctypes (some_amd/_lib.py).  hipcc cross-compiles without a GPU, so this also runs in the build container.

    python -m some_amd.build [--force]
"""
import concurrent.futures
import os
import pathlib
import shutil
import subprocess
import sys

HERE = pathlib.Path(__file__).resolve().parent
CSRC = HERE / 'csrc'
OBJ = CSRC / 'build'
LIB = HERE / 'libsome_amd.so'
SOURCES = ['api.hip', 'gemm.hip', 'gemm_f16x3.hip', 'rowops.hip', 'attention.hip', 'attention_f16x3.hip', 'dwconv.hip', 'logmel.hip', 'logmel_shift.hip', 'decode.hip', 'ingest.hip', 'train_ops.hip', 'train_attention.hip', 'train_attention_f16x3.hip', 'train_gemm16s.hip', 'train_api.hip', 'calibrate.hip']
HEADERS = [CSRC / 'internal.h', CSRC / 'fft_core.h', CSRC / 'split.h', CSRC / 'rms_core.h', HERE.parent / 'include' / 'some_amd.h']
ARCH = 'gfx950'
# per-source additions.  attention_f16x3.hip: without SLP vectorisation hipcc keeps the softmax's fp32 adds / multiplies scalar -
# packed v_pk_* instructions beside MFMAs cost more than the two plain ones they replace (MI355X_MICROARCH.md, cost table)
EXTRA_FLAGS = {'attention_f16x3.hip': ['-fno-slp-vectorize']}
FLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-function']


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found (need ROCm >= 7.0 to build libsome_amd.so)')
    return exe


def _stale(target: pathlib.Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(pathlib.Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> pathlib.Path:
    hipcc = _hipcc()
    OBJ.mkdir(parents=True, exist_ok=True)
    jobs = []
    for src in SOURCES:
        s, o = CSRC / src, OBJ / (src + '.o')
        if force or _stale(o, [s] + HEADERS):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(s.name, []) + os.environ.get('SOME_AMD_HIPCC_FLAGS', '').split() + ['-c', str(s), '-o', str(o)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {s.name}:\n{r.stdout}\n{r.stderr}')
        return s.name, r.stderr

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for name, err in ex.map(compile_one, jobs):
                if verbose:
                    print(f'[some_amd.build] compiled {name}' + (f'\n{err}' if err.strip() else ''))
    objs = [OBJ / (s + '.o') for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', str(LIB)] + [str(o) for o in objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        if verbose:
            print(f'[some_amd.build] linked {LIB}')
    return LIB


FASTCALL = HERE / '_fastcall.so'


def _load_module(path: pathlib.Path, name: str):
    """A module of this package by file, without importing the package (some_amd/__init__ pulls in torch)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build_fastcall(force: bool = False, verbose: bool = True) -> pathlib.Path:
    """some_amd/_fastcall.so: the generated CPython binding of the C ABI (some_amd/fastcall_gen.py) - plain C, gcc, no link dependency."""
    import sysconfig
    header = HERE.parent / 'include' / 'some_amd.h'
    deps = [HERE / '_lib.py', HERE / 'fastcall_gen.py', header]
    if not (force or _stale(FASTCALL, deps)):
        return FASTCALL
    gcc = shutil.which('gcc') or shutil.which('cc')
    if gcc is None:
        raise RuntimeError('gcc not found (needed for some_amd/_fastcall.so)')
    OBJ.mkdir(parents=True, exist_ok=True)
    src = OBJ / 'fastcall_gen.c'
    src.write_text(_load_module(HERE / 'fastcall_gen.py', '_some_amd_fastcall_gen').generate(_load_module(HERE / '_lib.py', '_some_amd_lib_sig').SYMBOLS))
    cmd = [gcc, '-O2', '-std=gnu11', '-shared', '-fPIC', '-Wall', '-Werror=implicit-function-declaration', f'-I{sysconfig.get_paths()["include"]}',
           f'-I{header.parent}', str(src), '-o', str(FASTCALL)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'gcc failed for {src.name}:\n{r.stdout}\n{r.stderr}')
    if verbose:
        print(f'[some_amd.build] built {FASTCALL}' + (f'\n{r.stderr}' if r.stderr.strip() else ''))
    return FASTCALL


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    build_fastcall(force='--force' in sys.argv)
