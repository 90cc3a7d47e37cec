import json, subprocess, sys
for chunks in (6, 7, 8):
    for pct in (2, 5, 10, 20):
        vals=[]
        for rep in range(2):
            out=subprocess.run([sys.executable,'bench.py','--only-config','cfg5f32','--cpu-seconds','0','--option','sync_chunks=%d'%chunks,'--option','sync_stagger=%d'%pct],capture_output=True,text=True).stdout.strip().splitlines()[-1]
            r=list(json.loads(out).values())[0]
            vals.append(r['host_to_host_kept_arrays_calls_per_sec'])
        print(chunks, pct, ['%.3g'%v for v in vals], flush=True)
