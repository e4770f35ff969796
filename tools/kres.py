"""Per-kernel resource usage of one csrc/*.hip file as hipcc reports it (registers, spills, scratch, LDS, waves per SIMD):
    python tools/kres.py keisei_amd/csrc/wgrad.hip [extra hipcc flags]"""
import re, subprocess, sys
src, extra = sys.argv[1], sys.argv[2:]
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage", *extra], capture_output=True, text=True)
cur = {}
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        m = re.search(r":\d+:\d+: remark: (.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for c in rows:
    name = subprocess.run(["c++filt", c["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    print(f"{name[:70]:70s} vgpr {c.get('VGPRs','?'):>4} agpr {c.get('AGPRs','?'):>4} spill {c.get('VGPR Spill','?'):>3} scratch {c.get('ScratchSize [bytes/lane]','?'):>5} "
          f"lds {c.get('LDS Size [bytes/block]','?'):>6} occ {c.get('Occupancy [waves/SIMD]','?')}")
if r.returncode: print(r.stderr[-3000:])
