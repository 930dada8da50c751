"""Registers, scratch and spills of every gfx950 kernel in the built library, read from the code objects' AMDGPU metadata (no GPU needed).
usage: python scripts/kernel_resources.py [path/to/libssfm_hip.so] [filter]
Why it exists (round 6): k_schur_gram carried 48 B of scratch for three rounds -- a select chain over eight scalar registers that the compiler had turned into a
per-lane scratch array at the head of every task -- and nothing but the ISA showed it.  tests/test_kernel_resources_cpu.py holds the hot kernels at zero."""
import os, re, struct, subprocess, sys, tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path, arch="gfx950"):
    """(offset, size) of every code object for `arch` inside the offload bundles embedded in a host library"""
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0: break
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            eo, es, ts = struct.unpack_from("<QQQ", data, off); off += 24
            triple = data[off:off + ts].decode(errors="replace"); off += ts
            if arch in triple and es > 0: out.append(data[i + eo:i + eo + es])
        pos = i + 24
    return out


def demangle_short(name):
    m = re.match(r"_ZN4ssfmL?(\d+)(.*)", name)
    if not m: return name
    n = int(m.group(1)); base = m.group(2)[:n]; rest = m.group(2)[n:]
    t = re.match(r"I((?:L[ib]\d+E)+)E", rest)
    if t: base += "<" + ", ".join(re.findall(r"L[ib](\d+)E", t.group(1))) + ">"
    return base


def kernels(path):
    res = {}
    for blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".o") as f:
            f.write(blob); f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for block in txt.split("\n  - .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", block) or [None, "0"])[1]
            name = g("name")
            res[name] = {"short": demangle_short(name), "agpr": int(block.split()[0]),                 # (the block starts with the value of .agpr_count)
                         "vgpr": int(g("vgpr_count")), "sgpr": int(g("sgpr_count")), "scratch": int(g("private_segment_fixed_size")),
                         "vgpr_spill": int(g("vgpr_spill_count")), "sgpr_spill": int(g("sgpr_spill_count")), "lds": int(g("group_segment_fixed_size")),
                         "dynamic_stack": g("uses_dynamic_stack") == "true"}
    return res


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spherical_sfm_amd", "libssfm_hip.so")
    flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    ks = kernels(lib)
    print("%-52s %5s %5s %5s %8s %6s %6s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "vspill", "sspill", "lds"))
    for name, k in sorted(ks.items(), key=lambda kv: kv[1]["short"]):
        if flt and flt not in k["short"]: continue
        print("%-52s %5d %5d %5d %8d %6d %6d %7d" % (k["short"][:52], k["vgpr"], k["agpr"], k["sgpr"], k["scratch"], k["vgpr_spill"], k["sgpr_spill"], k["lds"]))
    print(len(ks), "kernels")
