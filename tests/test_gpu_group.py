"""GPU tests of the round-3 entry points: the multi-device group (smafa_group_*), the one-pass-per-query launcher
(smafa_scan_each) and the per-call totals (smafa_last_call_stats) — all through the C ABI, against the oracle."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import smafa_amd
from smafa_amd import _lib, synth
from smafa_amd._lib import lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rows(a):
    return np.stack([a["query"], a["subject"], a["dist"]], axis=1).astype(np.uint32)


def _select_k(want, k):
    """oracle rows (every pair within the bound, ordered) -> the rows within each query's k-th smallest distance"""
    out = []
    for q in np.unique(want["query"]):
        r = want[want["query"] == q]
        kth = r["dist"][k - 1] if len(r) >= k else np.iinfo(np.uint32).max
        out.append(r[r["dist"] <= kth])
    return np.concatenate(out) if out else want[:0]


@pytest.mark.parametrize("alphabet", [smafa_amd.ALPHABET_NT, smafa_amd.ALPHABET_AA])
def test_group_rows_do_not_depend_on_the_number_of_handles(alphabet):
    """src/lib.rs:232-318 sharded: N in {1, 2, 3} handles (all on GPU 0) return byte-identical rows == the oracle's, for a
    fixed bound, best hit (k = 1) and k = 5; the store arrives in two appends"""
    n, L, nq = 40_000, 60, 1_001
    subj = synth.subjects(n, L, alphabet, seed=21)
    qry, _, _ = synth.queries(subj, nq, alphabet, seed=22, max_subs=8)
    D = 5
    want = oracle.scan_codes(subj, qry, D)
    everything = oracle.scan_codes(subj, qry[:64], L)  # every pair: the k-th modes without a bound, on 64 queries
    got = {}
    for ndev in (1, 2, 3):
        devs = [d % smafa_amd.device_count() for d in range(ndev)]  # [0, 0, 0] on a one-GPU box, [0, 1, 2] on a node
        g = smafa_amd.SubjectGroup(L, alphabet, devices=devs)
        assert len(g) == ndev
        g.push(subj[:25_000])
        g.push(subj[25_000:])
        fixed = g.scan(qry, max_divergence=D)
        assert _rows(fixed).tobytes() == _rows(want).tobytes()
        best = g.scan(qry[:64], max_num_hits=1)
        assert _rows(best).tobytes() == _rows(_select_k(everything, 1)).tobytes()
        k5 = g.scan(qry[:64], max_divergence=40, max_num_hits=5)
        w5 = _select_k(everything[everything["dist"] <= 40], 5)
        assert _rows(k5).tobytes() == _rows(w5).tobytes()
        # grow-and-retry: a buffer that is too small reports the size and the repeated call is answered
        small = g.scan(qry, max_divergence=D, cap=3)
        assert _rows(small).tobytes() == _rows(want).tobytes()
        got[ndev] = (fixed.tobytes(), best.tobytes(), k5.tobytes())
        # every member lives on ITS entry of `devices`, and its worker thread launched there: smafa_db_info().device, the HIP
        # device current on the launching thread, no launch off the handle's device (with more than one GPU visible the
        # members spread over them: the first real `--devices 0,1,..` run cannot silently put every replica on device 0)
        assert g.members() == [(d, d, 0) for d in devs], (devs, g.members())
        # every replica builds its own block index (smafa_group_build_index): the same rows from D + 1 probes per query
        infos = g.build_index(D)
        assert len(infos) == ndev and all(i["current"] == 1 and i["max_div_served"] == D for i in infos), infos
        assert g.scan(qry, max_divergence=D).tobytes() == fixed.tobytes()
        for m in range(ndev):
            info = _lib.IndexInfo()
            assert lib().smafa_index_info(lib().smafa_group_member(g._h, m), C.byref(info)) == 0
            assert info.probe_launches == 1, (m, info.probe_launches)  # 1 001 queries in ndev blocks of more than 64
        assert g.members() == [(d, d, 0) for d in devs], (devs, g.members())
        g.close()
    assert got[1] == got[2] == got[3]


def test_group_from_a_packed_store_file(tmp_path):
    n, L = 20_000, 60
    subj = synth.subjects(n, L, 0, seed=5)
    qry, _, _ = synth.queries(subj, 300, 0, seed=6, max_subs=6)
    one = smafa_amd.SubjectStore(L, 0, 0)
    one.push(subj)
    path = str(tmp_path / "s.packed")
    one.save(path)
    want = one.scan(qry, max_divergence=3)
    one.close()
    g = smafa_amd.SubjectGroup.load(path, devices=[0, 0])
    assert _rows(g.scan(qry, max_divergence=3)).tobytes() == _rows(want).tobytes()
    assert _rows(want).tobytes() == _rows(oracle.scan_codes(subj, qry, 3)).tobytes()
    g.close()


def test_scan_each_equals_the_batch_scan():
    """smafa_scan_each — tests/scan_each_worker.py, a process of its own because the device buffers come from torch, which
    has to initialise HIP before the library does"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "scan_each_worker.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "scan_each ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_last_call_stats_cover_every_scan_of_a_call():
    n, L = 200_000, 60
    subj = synth.subjects(n, L, 1, seed=41)
    near, _, _ = synth.queries(subj, 64, 1, seed=42, max_subs=4)
    far = synth.subjects(64, L, 1, seed=43, dup_frac=0.0)  # unrelated to every subject
    store = smafa_amd.SubjectStore(L, 1, 0)
    store.push(subj)
    rows = store.scan(np.concatenate([near, far]), max_num_hits=1)  # best hit, no bound: ladder + tightening path
    st = store.last_call_stats()
    assert st["scans"] >= 2 and st["launches"] >= st["scans"] and st["kernel_ms"] > 0
    assert len(np.unique(rows["query"])) == 128  # every query has a best hit
    store.scan(near, max_divergence=5)
    st1 = store.last_call_stats()
    assert st1["scans"] == 1 and st1["launches"] == 1
    store.close()
