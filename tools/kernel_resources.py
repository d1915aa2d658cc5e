#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch use of a built libdfgnn.so (from the code object's metadata notes).
usage: python tools/kernel_resources.py [pattern] [lib]"""
import re, subprocess, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "dense"
lib = sys.argv[2] if len(sys.argv) > 2 else "df-gnn_amd/libdfgnn.so"
import tempfile, os
tmp = tempfile.mkdtemp()
fb = os.path.join(tmp, "fb")
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fb], check=True)
blob = open(fb, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
offs = [m.start() for m in re.finditer(magic, blob)] + [len(blob)]
out = ""
for k in range(len(offs) - 1):   # one bundle per translation unit
    part, co = os.path.join(tmp, f"b{k}"), os.path.join(tmp, f"co{k}")
    open(part, "wb").write(blob[offs[k]:offs[k + 1]])
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--unbundle", f"--input={part}",
                    f"--output={co}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True, capture_output=True)
    out += subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
for blk in out.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name or not re.search(pat, name.group(1)): continue
    g = lambda k: (re.search(rf"\.{k}:\s+(\d+)", blk) or [0, "?"])[1]
    dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
    print(f"vgpr {g('vgpr_count'):>3} agpr {blk.split()[0]:>3} sgpr {g('sgpr_count'):>3} spill {g('vgpr_spill_count'):>3} "
          f"scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}  {dem[:110]}")
