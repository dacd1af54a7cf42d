"""Sample board power / shader clock with rocm-smi while a command runs:  python tools/power_sampler.py OUT.jsonl -- <cmd...>"""
import json, subprocess, sys, threading, time
out = sys.argv[1]; cmd = sys.argv[sys.argv.index("--") + 1:]
stop = False; rows = []
def sampler():
    t0 = time.time()
    while not stop:
        try:
            j = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout)
            c = j.get("card0", {})
            rows.append(dict(t=round(time.time() - t0, 2), power_w=float(c.get("Current Socket Graphics Package Power (W)", "nan")),
                             sclk=c.get("sclk clock speed:", "")))
        except Exception as e:
            rows.append(dict(t=round(time.time() - t0, 2), err=str(e)[:60]))
        time.sleep(0.2)
th = threading.Thread(target=sampler); th.start()
rc = subprocess.call(cmd)
stop = True; th.join()
with open(out, "w") as f:
    for r in rows: f.write(json.dumps(r) + "\n")
sys.exit(rc)
