"""bench.py's ALGORITHMIC bytes are compulsory HBM bytes: for every configuration of the committed PMC record
(profiles/pmc_traffic.json: rocprofv3 FETCH_SIZE / WRITE_SIZE passes of `bench.py --workload W --books B`, per book-step) no
kernel may claim more bytes per book-step than the counters saw move (+5 %: the passes are separate runs).  VERDICT r5 weak #4:
rounds 2 - 5 charged the wave-parallel decode's cache-resident lane-state record and block-start spills as HBM bytes, so C2's
"9.0 % of HBM" was really ~2.7 % - nothing checked it."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PMC = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
KEYS = sorted(k for k in PMC if not k.startswith("_"))
SETUP_KERNELS = ("k_wave_lists_rebuild",)  # one-off launches of a run, not step kernels: no per-step accounting


def test_every_pmc_configuration_carries_its_accounting():
    assert KEYS, "profiles/pmc_traffic.json holds no configuration"
    missing = [k for k in KEYS if "_accounting" not in PMC[k]]
    assert not missing, f"no accounting inputs (S, W4, ev, tr, new, spl, pipe) for {missing}: scripts/pmc_merge.py copies them from the passes' bench lines"


@pytest.mark.parametrize("key", KEYS)
def test_algorithmic_bytes_do_not_exceed_measured_traffic(key):
    cfg = PMC[key]
    acct = {k: v for k, v in cfg["_accounting"].items() if k != "note"}
    checked = 0
    for kernel, rec in cfg.items():
        if kernel.startswith("_") or kernel in SETUP_KERNELS or "hbm_bytes_per_book_step" not in rec:
            continue
        hbm, l2 = bench.algorithmic_bytes(kernel, **acct)
        measured = rec["hbm_bytes_per_book_step"]
        assert hbm > 0 and l2 >= 0
        assert hbm <= 1.05 * measured, (f"{key} {kernel}: {hbm:.0f} algorithmic B per book-step > 1.05 x {measured:.0f} B measured "
                                        f"(FETCH_SIZE x 2 + WRITE_SIZE): the accounting charges bytes that never reach HBM")
        checked += 1
    assert checked, f"{key}: no step kernel with measured traffic"


def test_c2_fraction_is_the_measured_one():
    """C2 (k_run_wave, fused): ~1.0 KB per book-step, not the 3.5 KB rounds 2 - 5 claimed."""
    acct = {k: v for k, v in PMC["C2/4096"]["_accounting"].items() if k != "note"}
    hbm, l2 = bench.algorithmic_bytes("k_run_wave", **acct)
    assert hbm < 1200.0 and l2 > 2000.0
