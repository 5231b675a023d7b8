"""Exposure of the shipped kernels to the packed-fp32 instruction form that was implicated in round 5's run-to-run
differences (profiles/r05_walker_packed_fp32_hazard.txt): per kernel of every object that is linked into
libpita_hip.so, the number of v_pk_{mul,add,fma}_f32 instructions, how many of them take an SGPR(-pair) source, how many
carry op_sel / op_sel_hi, and how many do both (the form of the failing sequence), next to the matrix instructions the
kernel issues beside them.  Read from the built objects, no GPU needed:

    python tools/packed_fp32_audit.py > profiles/r06_packed_fp32_exposure.txt

tests/test_kernel_resources.py::test_packed_fp32_exposure_of_shipped_kernels compares the committed listing with the
objects on disk; tests/test_hip_parity.py::test_default_path_full_batch_rerun is the run-time soak of the same kernels."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
PK = re.compile(r"\bv_pk_(mul|add|fma)_f32\b")


def disassemble(obj, tag="audit"):
    fb, co = f"/tmp/_pk_{tag}.fb", f"/tmp/_pk_{tag}.co"
    r = subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", obj, f"/tmp/_pk_{tag}.o"],
                       capture_output=True)
    if r.returncode != 0:  # an object without device code (abi.hip)
        return ""
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}", f"--output={co}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True, capture_output=True)
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout


def audit_object(obj):
    """{kernel name: dict(pk, sgpr, opsel, both, mfma16, mfma32)} for one host object with a bundled gfx950 code object."""
    asm = disassemble(obj, os.path.basename(obj).replace(".", "_"))
    out, name = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1)
            out[name] = dict(pk=0, sgpr=0, opsel=0, both=0, mfma16=0, mfma32=0)
            continue
        if name is None:
            continue
        r = out[name]
        if "v_mfma_f32_16x16x32" in line:
            r["mfma16"] += 1
        elif "v_mfma_f32_32x32x16" in line:
            r["mfma32"] += 1
        if PK.search(line):
            code = line.split("//")[0]
            s, o = " s[" in code or ", s[" in code, "op_sel" in code
            r["pk"] += 1
            r["sgpr"] += s
            r["opsel"] += o
            r["both"] += s and o
    return out


def demangle(names):
    if not names:
        return {}
    txt = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return {n: re.sub(r"\(.*", "", d).replace("void pita::", "") for n, d in zip(names, txt)}


def shipped_objects():
    sys.path.insert(0, ROOT)
    from pita_amd import build

    return [os.path.join(build.CSRC, s.replace(".hip", ".o")) for s in build.SOURCES]


def table():
    rows = []
    for obj in shipped_objects():
        k = audit_object(obj)
        names = demangle(list(k))
        tot = {f: sum(r[f] for r in k.values()) for f in ("pk", "sgpr", "opsel", "both")}
        rows.append((os.path.basename(obj), "(whole object)", tot))
        for n, r in sorted(k.items(), key=lambda kv: -kv[1]["both"]):
            if r["pk"]:
                rows.append((os.path.basename(obj), names[n], r))
    return rows


def main():
    print("Packed fp32 vector instructions in the kernels that ship in libpita_hip.so (tools/packed_fp32_audit.py)")
    print("pk = v_pk_{mul,add,fma}_f32; sgpr = with an SGPR(-pair) source; op_sel = with op_sel / op_sel_hi; both = the form of")
    print("the failing sequence of profiles/r05_walker_packed_fp32_hazard.txt; mfma16 / mfma32 = v_mfma_f32_16x16x32_* /")
    print("v_mfma_f32_32x32x16_* in the same kernel.  The kernel the fault was seen in (16x16x32 products, two waves per")
    print("SIMD, hipcc-generated packed code) is no longer in the library; none of the kernels below has shown a differing")
    print("bit in the full-batch rerun soak (tests/test_hip_parity.py::test_default_path_full_batch_rerun).")
    print()
    print(f"{'object':30s} {'pk':>6s} {'sgpr':>6s} {'op_sel':>6s} {'both':>6s} {'mfma16':>6s} {'mfma32':>6s}  kernel")
    for obj, name, r in table():
        print(f"{obj:30s} {r['pk']:6d} {r['sgpr']:6d} {r['opsel']:6d} {r['both']:6d} {r.get('mfma16', 0) or 0:6d} "
              f"{r.get('mfma32', 0) or 0:6d}  {name}")


if __name__ == "__main__":
    main()
