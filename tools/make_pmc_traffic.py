"""Builds profiles/pmc_traffic.json from the per-kernel PMC summaries of tools/profile.sh.

usage: make_pmc_traffic.py <tag> [<tag> ...]   (reads gpurun_out/<tag>_pmc_{fetch,write}.csv and
gpurun_out/calib_pmc_{FETCH_SIZE,WRITE_SIZE}.csv)

traffic[kernel] = HBM-side bytes per launch = FETCH_SIZE * kf + WRITE_SIZE * kw, where
kf, kw scale the raw KiB counters to bytes using the calibration copy (256 MiB each way).
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out')
CAL_BYTES = 256 * 2 ** 20

# bench.py kernel name -> substrings of the device kernels that implement it
MAP = {
    'gemm_logits_hinge': ['k_sc_hinge'],
    'gemm_dU_bits': ['k_sc_bits<false'],
    'gemm_dI_bits': ['k_sc_bits<true'],
    'scorer_prep': ['k_sc_prep'],
    'scorer_rows': ['k_sc_rows'],
    'scorer_tn_reduce': ['k_sc_tn_reduce'],
    'gemm_logits_nt': ['k_gemm_nt_areg'],
    'gemm_dU_nn': ['k_gemm_dma<true', 'k_gemm_f32<64, 64, 16, true, false>', 'k_gemm_f32<128, 128, 16, true, false>'],
    'gemm_dI_tn': ['k_gemm_dma<false', 'k_gemm_f32<64, 64, 16, false, false>', 'k_gemm_f32<128, 128, 16, false, false>'],
    'splitk_reduce': ['k_splitk_reduce'],
    'loss_mw': ['k_loss_margin'],
    'gather_onehot': ['k_gather_onehot'],
    'gather_mulhot': ['k_gather_mulhot'],
    'lookup_multi': ['k_lookup_multi'],
    'sparse_apply_window': ['k_sparse_win'],
    'sparse_apply_small': ['k_sparse_onepass'],
    'sparse_finish': ['k_sparse_finish'],
    'sparse_apply_runs': ['k_run_apply'],
    'runs_extract': ['k_runs_extract'],
    'bag_expand_compact': ['k_bag_expand_compact'],
    'radix_scatter': ['k_rs_scatter'],
    'radix_hist': ['k_rs_hist'],
    'bag_expand_heads': ['k_bag_expand_heads'],
}


def read(path, col):
    d = {}
    if not os.path.exists(path):
        return d
    for r in csv.DictReader(open(path)):
        d[r['kernel']] = (float(r[col]), int(r['dispatches']))
    return d


def calib(counter):
    d = read(os.path.join(OUT, 'calib_pmc_%s.csv' % counter), 'max')   # largest dispatch = the 256 MiB copy
    best = None
    for k, (v, n) in d.items():
        if 'copy' in k.lower() or 'elementwise' in k.lower():
            if best is None or v > best:
                best = v
    return (CAL_BYTES / (best * 1024.0)) if best else None


def main():
    kf, kw = calib('FETCH_SIZE'), calib('WRITE_SIZE')
    res = {'_calibration': {'fetch_scale': kf, 'write_scale': kw,
                            'note': 'bytes = raw KiB * 1024 * scale; scale from a 256 MiB streaming copy'}}
    kf = kf or 2.0
    kw = kw or 1.0
    for tag in sys.argv[1:]:
        f = read(os.path.join(OUT, '%s_pmc_fetch.csv' % tag), 'mean_FETCH_SIZE')
        w = read(os.path.join(OUT, '%s_pmc_write.csv' % tag), 'mean_WRITE_SIZE')
        entry = {}
        for name, subs in MAP.items():
            fb = wb = 0.0
            hit = False
            for k, (v, n) in f.items():
                if any(s in k for s in subs):
                    fb += v * 1024.0 * kf
                    hit = True
            for k, (v, n) in w.items():
                if any(s in k for s in subs):
                    wb += v * 1024.0 * kw
            if hit:
                entry[name] = {'traffic_bytes': fb + wb, 'fetch_bytes': fb, 'write_bytes': wb}
        res[tag] = entry
    json.dump(res, open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'), 'w'), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True)[:3000])


if __name__ == '__main__':
    main()
