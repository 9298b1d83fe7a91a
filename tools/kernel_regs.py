"""VGPR / scratch / LDS of the kernels in a built translation unit (default: curve_bls12_381.o), read from the code object's
notes: `python tools/kernel_regs.py [object] [name filter ...]`."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
obj = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".o") else os.path.join(ROOT, "poly_commit_amd/csrc/_obj/curve_bls12_381.o")
filters = [a for a in sys.argv[1:] if not a.endswith(".o")]
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
for blk in notes.split("- .agpr_count")[1:]:
    g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk)
    name = g("name").group(1)
    try: name = subprocess.check_output([f"{LLVM}/llvm-cxxfilt", name], text=True).strip()
    except Exception: pass
    if filters and not any(f in name for f in filters): continue
    print(f"{name[:110]:110s} vgpr {g('vgpr_count').group(1):>4s} sgpr {g('sgpr_count').group(1):>4s} scratch {g('private_segment_fixed_size').group(1):>5s} lds {g('group_segment_fixed_size').group(1):>6s}")
