#!/usr/bin/env python
"""Build a VARIANT of libsome_amd.so for kernel A/B experiments: the named sources are recompiled with extra -D flags
into tools/_bin/variants/<name>/ and linked with the stock objects of everything else; select it at run time with
SOME_AMD_LIBRARY=tools/_bin/variants/<name>/libsome_amd.so (travels to the GPU box with the snapshot).

    python tools/build_variant.py prio1 attention_f16x3.hip -DATTN_SETPRIO=1
"""
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from some_amd import build as B  # noqa: E402

name, rest = sys.argv[1], sys.argv[2:]
srcs = [a for a in rest if a.endswith('.hip')]
flags = [a for a in rest if not a.endswith('.hip')]
B.build(verbose=False)
out = ROOT / 'tools' / '_bin' / 'variants' / name
out.mkdir(parents=True, exist_ok=True)
objs = []
for s in B.SOURCES:
    if s in srcs:
        o = out / (s + '.o')
        subprocess.check_call([B._hipcc()] + B.FLAGS + B.EXTRA_FLAGS.get(s, []) + flags + ['-c', str(B.CSRC / s), '-o', str(o)])
    else:
        o = B.OBJ / (s + '.o')
    objs.append(str(o))
lib = out / 'libsome_amd.so'
subprocess.check_call([B._hipcc(), '-shared', '-fPIC', f'--offload-arch={B.ARCH}', '-o', str(lib)] + objs)
print(lib)
