"""Lab only: writes band_kernels2_stamped.h, a copy of the product header whose k_band_chol_v2 records s_memtime stamps of
workgroup 0 in LDS (before / after the panel phase, after barrier 1, after the role's work) and dumps them at the end."""
import os, re
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "..", "..", "spherical_sfm_amd", "csrc", "band_kernels2.h")).read()
old_sig = "int* __restrict__ flags = nullptr, int seq = 0) {\n    constexpr int BB = DC * DC;\n    extern __shared__"
assert old_sig in src
src = src.replace(old_sig, "int* __restrict__ flags = nullptr, int seq = 0, long long* __restrict__ dbg = nullptr) {\n    constexpr int BB = DC * DC;\n    extern __shared__", 1)
src = src.replace("    int* sPairs = reinterpret_cast<int*>(sD + BB);",
                  "    int* sPairs = reinterpret_cast<int*>(sD + BB);\n    long long* sStamp = reinterpret_cast<long long*>(sPairs + b * (b + 1) / 2 + 2 + ((b * (b + 1) / 2) & 1));\n"
                  "#define STAMP(j_, k_) do { if (blockIdx.x == 0 && lane == 0) sStamp[((size_t)((j_) - r0) * nw + wave) * 4 + (k_)] = (long long)__builtin_readcyclecounter(); } while (0)", 1)
# stamps around phaseB / barriers in every role loop
src = src.replace("            phaseB(j, jm, nb);\n", "            STAMP(j, 0); phaseB(j, jm, nb); STAMP(j, 1);\n")
src = src.replace("            phaseB((j_), jm, nb);                                                                                     \\\n",
                  "            STAMP((j_), 0); phaseB((j_), jm, nb); STAMP((j_), 1);                                                     \\\n")
# after barrier 1 and before barrier 2: the first lds_barrier after phaseB and the last in the loop body
lines = src.split("\n"); out = []; state = 0
for ln in lines:
    if "STAMP(" in ln and "phaseB" in ln: state = 1
    if "lds_barrier();" in ln and "define" not in ln and state in (1, 2):
        jvar = "(j_)" if ln.rstrip().endswith("\\") else "j"
        if state == 1:
            out.append(ln.replace("lds_barrier();", f"lds_barrier(); STAMP({jvar}, 2);")); state = 2; continue
        else:
            out.append(ln.replace("lds_barrier();", f"STAMP({jvar}, 3); lds_barrier();")); state = 0; continue
    out.append(ln)
src = "\n".join(out)
# dump at the end of the kernel: find the end of k_band_chol_v2 (the closing of the writer branch)
marker = "    if (sig >= 0) {                                        // this segment's share of its separator is in global memory"
assert marker in src
src = src.replace(marker, "    __syncthreads();\n    if (dbg && blockIdx.x == 0) for (int e = tid; e < (r1 - r0) * nw * 4; e += nt) dbg[e] = sStamp[e];\n" + marker, 1)
open(os.path.join(here, "band_kernels2_stamped.h"), "w").write(src)
print("stamped header written; STAMP count", src.count("STAMP("))
