#!/usr/bin/env python3
"""Per-layer HBM traffic of the last profiled step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
against the algorithmic bytes (exact mode, B=16, 3-class 512x512).  usage: pmc_layers.py FETCH_CSV WRITE_CSV"""
import csv, sys
def per_dispatch(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter and 'weight_' not in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return rows
f = per_dispatch(sys.argv[1], 'FETCH_SIZE'); w = per_dispatch(sys.argv[2], 'WRITE_SIZE')
names = ["convert","c00.1","c00.2","c10.1","c10.2","c20.1","c20.2","c30.1","c30.2","c40.1","c40.2","up3","c31.1","c31.2","up2","c22.1","c22.2","up1","c13.1","c13.2","up0","c04.1","c04.2+head"]
B, P = 16, 2
def conv(px, cin, cout, pool=False, head=False): return px*P*2*(cin + (0 if head else cout)) + (px/4*P*2*cout if pool else 0)
lv = [262144*B/(4**l) for l in range(5)]; nb = [32, 64, 128, 256, 512]
alg = [lv[0]*(12+32)]
for l in range(5):
    alg.append(conv(lv[l], 8 if l == 0 else nb[l-1], nb[l])); alg.append(conv(lv[l], nb[l], nb[l], pool=l < 4))
for l in (3, 2, 1, 0):
    alg.append(lv[l]*P*2*nb[l+1]*1.25); alg.append(conv(lv[l], nb[l]+nb[l+1], nb[l])); alg.append(conv(lv[l], nb[l], nb[l], head=(l == 0)))
n = len(names); last = len(f) - n; tr = tw = ta = 0
for i, nm in enumerate(names):
    fr = float(f[last+i]['Counter_Value'])*2*1024; wr = float(w[last+i]['Counter_Value'])*1024
    tr += fr; tw += wr; ta += alg[i]
    print(f"{nm:12s} read {fr/1e6:8.1f} MB  write {wr/1e6:8.1f} MB  total {(fr+wr)/1e6:8.1f}  alg {alg[i]/1e6:8.1f}  ratio {(fr+wr)/alg[i]:.2f}")
print(f"step total: read {tr/1e9:.2f} GB write {tw/1e9:.2f} GB  algorithmic {ta/1e9:.2f} GB  ratio {(tr+tw)/ta:.2f}")
