#!/usr/bin/env python3
"""Per-launch HBM traffic of the last profiled step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units,
reads = 2 x FETCH_SIZE: the gfx950 wide-read correction of MI355X_MICROARCH.md §HBM) against each launch's algorithmic
bytes.  Launch labels and algorithmic bytes come from the engine itself (unetpp_profile_name / _work of one forward of
the same workload on this box), so the table follows whatever the op list is.
usage: pmc_layers.py FETCH_CSV WRITE_CSV [exact|fast] [B H W C]"""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_dispatch(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter and 'weight_' not in r['Kernel_Name']
            and 'tapw_pack' not in r['Kernel_Name'] and 'conv0_pack' not in r['Kernel_Name'] and 'convt_' not in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return rows


def main():
    f = per_dispatch(sys.argv[1], 'FETCH_SIZE'); w = per_dispatch(sys.argv[2], 'WRITE_SIZE')
    prec = sys.argv[3] if len(sys.argv) > 3 else "exact"
    B, H, W, C = (int(v) for v in sys.argv[4:8]) if len(sys.argv) > 7 else (16, 512, 512, 3)
    import torch
    from unet_amd import synthetic as syn
    from unet_amd.nested_unet import NestedUNet
    m = NestedUNet(C, deep_supervision=(C == 3), precision=prec, max_batch=B, max_hw=(H, W)).to("cuda:0")
    m.load_state_dict(syn.make_state_dict(C, 3, C == 3, 2))
    x = torch.from_numpy(syn.frames_to_chw_f32(syn.make_frames_u8(B, H, W, "smooth", 1234))).cuda()
    m.segment(x); m.profile(True); m.segment(x); torch.cuda.synchronize()
    recs = m.profile_read()
    n = len(recs); last = len(f) - n; tr = tw = ta = 0.0
    assert last >= 0 and len(w) == len(f), (len(f), len(w), n)
    for i, (name, _, _, alg) in enumerate(recs):
        kernel = name.split("|")[-1].split("<")[0]
        assert kernel in f[last + i]['Kernel_Name'], (name, f[last + i]['Kernel_Name'])     # same launch order
        fr = float(f[last + i]['Counter_Value']) * 2 * 1024; wr = float(w[last + i]['Counter_Value']) * 1024
        tr += fr; tw += wr; ta += alg
        print(f"{name.split('|')[0]:44s} read {fr/1e6:8.1f} MB  write {wr/1e6:8.1f} MB  total {(fr+wr)/1e6:8.1f}  alg {alg/1e6:8.1f}  ratio {(fr+wr)/alg:.2f}")
    print(f"step total: read {tr/1e9:.2f} GB write {tw/1e9:.2f} GB  algorithmic {ta/1e9:.2f} GB  ratio {(tr+tw)/ta:.2f}")


if __name__ == "__main__":
    main()
