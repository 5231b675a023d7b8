"""Register / scratch / LDS usage of every kernel in the built objects (development aid):
python tools/kernel_resources.py [substring]"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for obj in sorted(glob.glob(os.path.join(ROOT, "pita_amd", "csrc", "*.o"))):
    # the device code object is bundled inside the host object
    tmp, fb = "/tmp/_kr.co", "/tmp/_kr.fatbin"
    r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", obj], capture_output=True)
    if r.returncode != 0:
        continue
    r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                        f"--output={tmp}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True)
    if r.returncode != 0:
        continue
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", tmp], capture_output=True, text=True).stdout
    for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", txt, re.S):
        ag, lds, name, scr, sg, vg = m.groups()
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem)
        if pat in dem:
            print(f"{os.path.basename(obj):24s} {dem:60s} vgpr {vg:>4s} agpr {ag:>4s} sgpr {sg:>4s} scratch {scr:>5s} lds {lds:>6s}")
