"""Dev: one row of profiles/r05_pricing_repro.md — for the box this runs on, the event-pair time of the step's largest
families (bench.py priced_step, das_prof_* event pairs) next to the kernel time rocprofv3 saw for the same kernels.
usage: python tools/dev/pricing_repro.py <bench line .json> <rocprof .db> [label]
The rocprof run covers warm-up + timed + pricing steps of the same command; its per-step kernel time is the family's
total over the number of steps (one assign_targets_kernel launch per step)."""
import json
import re
import socket
import sqlite3
import sys

FAMS = {   # priced family tag -> kernels (regex on the trace's kernel name)
    'bn_apply_kernel': r'bn_apply_kernel|bn_apply_stream_kernel|bn_finalize_kernel|bn_fold_kernel|bn_dual_apply_kernel|'
                       r'bn_relu_add3_fwd_kernel|upmerge_fwd_kernel|upstats_lowres_kernel',
    'bn_bwd_apply_dz_kernel': r'bn_bwd_apply_dz_kernel|bn_bwd_apply_dz_stream_kernel',
    'conv1x1_stream_kernel': r'conv1x1_stream_kernel',
    'conv_wgrad_pp_kernel': r'conv_wgrad_pp_kernel|wgrad_reduce_kernel<.*AccMap256',
    'conv_wgrad_kernel<bf16>': r'conv_wgrad_kernel<|wgrad_reduce_kernel<.*AccMap128|conv_wgrad_c64_kernel|wgrad_c64_reduce_kernel',
}


def main(line_path, db, label=''):
    with open(line_path) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    pr = line['priced_step']
    c = sqlite3.connect(db)
    rows = c.execute('select name, count(*), sum(end-start) from kernels group by name').fetchall()
    steps = sum(n for name, n, _ in rows if 'assign_targets_kernel' in name)
    out = [f'box {socket.gethostname()} {label}: {line["ms_per_step"]} ms/step timed, pass {pr["step_ms_this_pass"]} ms '
           f'(median of the passes; min {pr.get("step_ms_this_pass_min")}), families {pr["families_ms_sum"]} ms, unreliable={pr.get("unreliable")}, '
           f'attempts={pr.get("attempts")}, headline {line["roofline"]["kernel"]} frac {line["roofline"]["frac"]}; '
           f'rocprof run: {steps} steps']
    out.append('| family | event pairs, median (ms/step) | event pairs, min | rocprof kernel time (ms/step) | events (median) / kernels |')
    out.append('|---|---|---|---|---|')
    for tag, rx in FAMS.items():
        ev = pr['families_ms'].get(tag)
        med = pr.get('families_ms_min', {}).get(tag)
        kt = sum(t for name, _, t in rows if re.search(rx, name)) / 1e6 / max(steps, 1)
        if ev is None:
            continue
        out.append(f'| {tag} | {ev:.3f} | {med if med is None else round(med, 3)} | {kt:.3f} | {ev / kt if kt else float("nan"):.3f} |')
    print('\n'.join(out))


if __name__ == '__main__':
    main(*sys.argv[1:4])
