"""Instruction histogram of a kernel's innermost loops, from the BUILT object, priced at the measured issue costs
(profiles/r03_isa_rates.txt) -- so that an "instruction floor" quoted in DESIGN.md can be recomputed by anyone.

    python tools/kernel_histogram.py [object] [kernel-substring]      (defaults: egnn_kernel.o, the LJ13 fused sampler)

A loop = the address range of a backward branch.  For every loop that contains matrix instructions: counts per
instruction class and the SIMD cycles they cost at two waves per SIMD (cycles per wave-instruction per SIMD, i.e. the
per-wave cost when two waves share the issue port: plain vector 2.4, half-rate 4.45, transcendental 8.3; a 16-bit MFMA
next to saturated vector issue 24 of its 32 cycles, the f32 MFMA 64 -- it runs on the vector datapath; LDS and scalar
instructions issue beside the vector pipe and are listed unpriced)."""
import collections, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "pita_amd", "csrc", "egnn_kernel.o")
pat = sys.argv[2] if len(sys.argv) > 2 else "egnn_kernel<13, 3, 7, 4, 2, true, 2>"

HALF = ("v_cvt_pk", "v_cvt_pkrtz", "v_fma_mix", "v_pk_", "v_perm_b32", "v_permlane", "v_mov_b32_dpp", "v_add_f32_dpp")
TRANS = ("v_exp_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag")
COST = {"plain vector": 2.4, "half-rate vector (cvt_pk / fma_mix / pk_* / perm / dpp)": 4.45, "transcendental": 8.3,
        "16-bit MFMA 32x32x16": 24.0, "f32 MFMA 32x32x2": 64.0}


def classify(mn, ops):
    if mn.startswith("v_mfma"):
        return "f32 MFMA 32x32x2" if "x2_f32" in mn or "x2f32" in mn else "16-bit MFMA 32x32x16"
    if mn.startswith("v_accvgpr"):
        return "plain vector"
    if mn in TRANS or mn.startswith(TRANS):
        return "transcendental"
    if mn.startswith(HALF) or "dpp" in ops or "row_" in ops or "quad_perm" in ops:
        return "half-rate vector (cvt_pk / fma_mix / pk_* / perm / dpp)"
    if mn.startswith("v_"):
        return "plain vector"
    if mn.startswith("ds_"):
        return "LDS"
    if mn.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vector memory"
    if mn == "s_nop":
        return "s_nop"
    if mn == "s_waitcnt":
        return "s_waitcnt"
    return "scalar / branch"


fb, co = "/tmp/_kh.fb", "/tmp/_kh.co"
subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", obj, "/tmp/_kh.o"], check=True, capture_output=True)
subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}", f"--output={co}",
                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True, capture_output=True)
dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--demangle", co], check=True, capture_output=True, text=True).stdout
m = re.search(r"^[0-9a-f]+ <([^\n]*" + re.escape(pat) + r"[^\n]*)>:\n(.*?)(?=^\s*$|\Z)", dis, re.S | re.M)
if not m:
    sys.exit(f"kernel matching '{pat}' not found in {obj}")
name, body = m.group(1), m.group(2)
ins = []
for ln in body.splitlines():
    mm = re.match(r"\s+(\S+)\s*(.*?)\s*// ([0-9A-Fa-f]+):", ln)
    if mm:
        ins.append((int(mm.group(3), 16), mm.group(1), mm.group(2)))
addr = [a for a, _, _ in ins]
loops = []
for k, (a, mn, ops) in enumerate(ins):
    if mn.startswith("s_cbranch") or mn == "s_branch":
        off = int(ops.split()[0])
        if off >= 32768:
            tgt = a + 4 + 4 * (off - 65536)
            lo = next(i for i, x in enumerate(addr) if x >= tgt)
            loops.append((lo, k))
# innermost loops WITH matrix instructions: drop any such loop that contains another one (short wait / copy loops inside
# an item loop do not hide it)
loops = sorted(set(loops))
has_mfma = lambda l: any(mn.startswith("v_mfma") for _, mn, _ in ins[l[0]:l[1] + 1])
mloops = [l for l in loops if has_mfma(l)]
inner = [l for l in mloops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in mloops)]
print(f"{os.path.relpath(obj, ROOT)}: {name.split('(')[0]}")
print(f"{len(ins)} instructions, {len(loops)} loops, {len(inner)} innermost with matrix instructions\n")
for lo, hi in inner:
    seg = ins[lo:hi + 1]
    h = collections.Counter(classify(mn, ops) for _, mn, ops in seg)
    if not any(k.endswith(("32x32x16", "32x32x2")) for k in h):
        continue
    detail = collections.Counter(mn for _, mn, _ in seg)
    print(f"loop at 0x{seg[0][0]:x} .. 0x{seg[-1][0]:x}: {len(seg)} instructions")
    tot = 0.0
    for cls in ("plain vector", "half-rate vector (cvt_pk / fma_mix / pk_* / perm / dpp)", "transcendental",
                "16-bit MFMA 32x32x16", "f32 MFMA 32x32x2", "LDS", "vector memory", "scalar / branch", "s_waitcnt", "s_nop"):
        n = h.get(cls, 0)
        c = COST.get(cls)
        tot += n * c if c else 0.0
        print(f"    {cls:58s} {n:5d}" + (f"  x {c:5.2f} = {n * c:8.0f} cycles" if c else ""))
    print(f"    {'priced issue cycles per iteration and wave (two waves per SIMD)':58s}        {tot:8.0f}")
    top = ", ".join(f"{k} {v}" for k, v in detail.most_common(14))
    print(f"    most frequent: {top}\n")
