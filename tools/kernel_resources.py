"""Register / scratch usage of every kernel in csrc/usvmpc.hip (development aid):
python tools/kernel_resources.py [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpc_collisionavoidance_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I" + os.path.join(CSRC, "gfx950"),
       "-I" + CSRC, "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/usv_res.o", os.path.join(CSRC, "usvmpc.hip")] + sys.argv[1:]
out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[bytes/lane\])?: (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("void ", "").replace("usv::", "")
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
print("%-58s %5s %5s %6s %7s %4s" % ("kernel", "VGPR", "AGPR", "spill", "scratch", "occ"))
for k, r in rows.items():
    print("%-58s %5s %5s %6s %7s %4s" % (k[:58], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize"), r.get("Occupancy")))
