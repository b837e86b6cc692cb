"""CPU: the measurement plumbing of bench.py that needs no GPU -- which rocprofv3 --pmc summary a workload's roofline block reads, that
counters taken from other kernel sources than the built library's are refused, and that the summaries committed under profiles/ for
this round belong to the kernel sources in the tree (a kernel edit without a new tools/pmc_bench.sh run would otherwise ship a bench
line whose `traffic` is silently null)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_newest_pmc_summary_per_workload_and_its_hash():
    import bench
    from rdfc_gan_amd import build
    for suffix in ('', '_bf16', '_nlspn', '_480x640'):
        tab, meta, fname = bench.pmc_table(suffix)
        assert tab is not None, f'no profiles/rNN_pmc_traffic{suffix}.json'
        rounds = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith(f'_pmc_traffic{suffix}.json'))
        assert os.path.basename(fname) == rounds[-1], (fname, rounds)
        assert meta.get('csrc_sha16') == build.source_hash(), (f'{fname} was measured on kernel sources {meta.get("csrc_sha16")}, the tree holds '
                                                               f'{build.source_hash()}: re-run tools/final_round.sh (or tools/pmc_bench.sh) on the GPU box')
        assert meta.get('fabric_gb_per_step', 0) > 10 and abs(sum(meta['fabric_gb_by_category'].values()) - meta['fabric_gb_per_step']) < 0.1


def test_stale_counters_are_refused_and_fresh_ones_fill_the_block():
    import bench
    tab, meta, fname = bench.pmc_table('')
    top = 'conv3x3_glds_kernel<128, 2, 10>'
    assert any(top + '(' in k for k in tab), 'the dominant kernel of the headline has no counters in the summary'
    roof = dict(traffic=None, avg_launch_us=120.0)
    assert bench.pmc_for_kernel(roof, top, '', 'not-the-hash') is None
    assert roof['traffic'] is None and 'stale counters refused' in roof['traffic_note']
    roof = dict(traffic=None, avg_launch_us=120.0)
    m = bench.pmc_for_kernel(roof, top, '', meta['csrc_sha16'])
    assert m is not None and roof['traffic'] > 5e7 and 0.05 < roof['mfma_busy']['frac_of_simd_cycles'] < 1.0
    tag = os.path.basename(fname).split('_')[0]                       # the bench line of the same round's artefacts
    line = json.loads(open(os.path.join(ROOT, 'profiles', f'{tag}_bench.json')).read().strip().splitlines()[-1])
    assert line['roofline']['traffic'] == roof['traffic'] and abs(line['fabric_gb_per_step'] - m['fabric_gb_per_step']) < 1e-6
