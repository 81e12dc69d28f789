#!/usr/bin/env python3
"""A/B libraries for one kernel file (or several: stem1+stem2): tools/build_variants.py <source stem> name=-DFLAG[,-DFLAG2] ...
Each variant recompiles only csrc/<stem>.hip with the extra flags and links it with the objects of the shipped build
into vietvoice-tts_amd/build/variants/libvvtts_<name>.so (they travel to the GPU box with the snapshot; git-ignored)."""
import importlib.util, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("vv_build_ext", os.path.join(ROOT, "vietvoice-tts_amd", "build_ext.py"))
be = importlib.util.module_from_spec(spec); spec.loader.exec_module(be)
be.build()
stems = sys.argv[1].split("+")          # several kernel files may take the flags: vv_gemm+vv_elementwise
out_dir = os.path.join(be.HERE, "build", "variants")
os.makedirs(out_dir, exist_ok=True)
for item in sys.argv[2:]:
    name, _, flags = item.partition("=")
    built = {}
    for stem in stems:
        obj = os.path.join(out_dir, f"{stem}_{name}.o")
        cmd = [be._hipcc()] + be.FLAGS + [f for f in flags.split(",") if f] + ["-c", os.path.join(be.CSRC, stem + ".hip"), "-o", obj]
        subprocess.run(cmd, check=True)
        built[stem] = obj
    objs = [built.get(s, os.path.join(be.OBJ, s + ".o")) for s in be.SOURCES]
    lib = os.path.join(out_dir, f"libvvtts_{name}.so")
    subprocess.run([be._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + be.LINK_FLAGS + ["-o", lib] + objs, check=True)
    print(lib)
