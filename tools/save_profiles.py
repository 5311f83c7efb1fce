"""Copy the summaries of a tools/profile_round.sh run from gpurun_out/ into profiles/ (tracked) and append the corrected
HBM traffic of the QP kernel to profiles/pmc_traffic.json.   usage: python tools/save_profiles.py <tag> <workload>"""
import csv, json, os, shutil, sys
tag, workload = sys.argv[1], sys.argv[2]
src, dst = "gpurun_out/prof_" + tag, "profiles"
shutil.copy(src + "/trace/t_kernel_stats.csv", "%s/%s_kernel_stats.csv" % (dst, tag))
shutil.copy(src + "/pmc_summary.csv", "%s/%s_pmc_summary.csv" % (dst, tag))
# kernel trace: keep the columns that matter (name, start, end, registers, scratch, LDS, grid)
rows = list(csv.DictReader(open(src + "/trace/t_kernel_trace.csv")))
keep = [c for c in ("Kernel_Name", "Start_Timestamp", "End_Timestamp", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size",
                    "LDS_Block_Size", "Workgroup_Size", "Grid_Size") if c in rows[0]]
with open("%s/%s_kernel_trace.csv" % (dst, tag), "w", newline="") as f:
    w = csv.DictWriter(f, keep); w.writeheader()
    for r in rows:
        if "usv_" in r["Kernel_Name"]:
            w.writerow({k: r[k] for k in keep})
line = [l for l in open(src + "/trace.log").read().splitlines() if l.startswith("{")][-1]
open("%s/%s_bench.json" % (dst, tag), "w").write(line + "\n")
b = json.loads(line)
f = wv = n = None
kname = None
cond_N = b["config"].get("qp_solver_cond_N")
cond_N = int(cond_N) if isinstance(cond_N, int) and cond_N != b["config"]["horizon"] else 0   # (partially condensed solve: kernel usv_qp_cond)
KERN = "qp_cond" if cond_N else "qp_rti"
fu = None   # the follow-up launch of a hand-over (usv_qp_resume): its bytes and its time belong to the same solve
for r in csv.reader(open(src + "/pmc_summary.csv")):
    if KERN in r[0]:
        kname, n, f, wv = r[0], int(r[1]), float(r[2]), float(r[3])
    if KERN == "qp_rti" and "qp_resume" in r[0]:   # (behind the launch, usv_qp_resume, and beside it, usv_qp_resume_co: the bytes of both belong to the solve)
        fu = (int(r[1]), (fu[1] if fu else 0.0) + float(r[2]), (fu[2] if fu else 0.0) + float(r[3]))
ms = None
ms_fu = 0.0
for r in csv.DictReader(open(src + "/trace/t_kernel_stats.csv")):
    if KERN in r["Name"]:
        ms = float(r["AverageNs"]) / 1e6
    if KERN == "qp_rti" and "qp_resume" in r["Name"] and "qp_resume_co" not in r["Name"]:   # (the kernel beside the launch overlaps it: its span is not added)
        ms_fu = float(r["AverageNs"]) / 1e6
if fu is not None:
    f, wv, ms = f + fu[1], wv + fu[2], ms + ms_fu
    kname += " + usv_qp_resume (+ usv_qp_resume_co beside the launch)"
tot = (2 * f + wv) * 1024
pj = dst + "/pmc_traffic.json"
J = json.load(open(pj))
J = [e for e in J if e.get("round") != tag]
cfgw = b["config"]
import subprocess
try:
    head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
    if subprocess.check_output(["git", "status", "--porcelain", "--", "mpc_collisionavoidance_amd/csrc"], text=True).strip():
        head += "+uncommitted kernel changes"
except Exception:
    head = None
J.append({"round": tag, "workload": workload, "lib_sha256": cfgw.get("lib_sha256"), "git_head": head, "moving": "moving" in cfgw.get("workload", "") and " static " not in cfgw.get("workload", ""), "kernel": kname.replace("void ", ""), "model": cfgw["ocp"], "N": cfgw["horizon"], "K": cfgw["obstacles"],
          "batch": cfgw["instances_per_gpu"], "cond_N": cond_N, "fetch_size_KiB_raw": f, "write_size_KiB_raw": wv, "hbm_bytes_per_launch": tot,
          "correction": "FETCH_SIZE x2, WRITE_SIZE x1; re-calibrated in this round on the [stage][group][plane][16 lanes] layout with "
                        "usv_calib_stream: 524288 KiB read -> FETCH_SIZE 262166 KiB, 2097152 KiB -> 1048612 KiB (factor 0.5000), "
                        "8192 KiB written -> WRITE_SIZE 8192 KiB (profiles/r02_a_calibration.txt)",
          "derived": "%.0f GB per launch at %.1f ms average launch duration (rocprofv3 kernel stats, %d launches) = %.2f TB/s = %.0f %% of the 8 TB/s peak; "
                     "%.2f MB per solve = %.0f x the algorithmic %d B" % (tot / 1e9, ms, n, tot / ms / 1e9, 100 * tot / ms / 1e9 / 8, tot / cfgw["instances_per_gpu"] / 1e6,
                                                                        tot / cfgw["instances_per_gpu"] / b["roofline"]["algorithmic_bytes_per_solve"], b["roofline"]["algorithmic_bytes_per_solve"]),
          "source": "profiles/%s_pmc_summary.csv, profiles/%s_kernel_stats.csv" % (tag, tag)})
json.dump(J, open(pj, "w"), indent=1)
print(tag, "%.0f GB/launch" % (tot / 1e9), "%.1f ms" % ms, b["value"])
