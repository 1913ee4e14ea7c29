"""The device half of the structure build (mc_slam_amd/csrc/vba_structure.h: record orders, keyframe segments, per-pair item
lists -- g2o's buildStructure, block_solver.hpp:143-295) against a plain numpy reconstruction from the same raw arrays."""
import ctypes as C

import numpy as np
import pytest

from mc_slam_amd import abi, synth, backend
from test_gpu_parity import _shuffled

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    b = backend.LocalBA(0, hooks=True)
    yield b
    b.close()


def _buf(ba, name, dtype, count, offset=0):
    a = np.zeros(count, dtype=dtype)
    rc = ba.lib.vba_debug_copy(ba.h, ba.lib.vba_debug_buf_id(name.encode()), C.c_uint64(offset), a.ctypes.data_as(C.c_void_p), C.c_uint64(a.nbytes))
    assert rc == 0, name
    return a


def _expected(p):
    """numpy twin: first keyframe of every track, the record orders and the item sets per off-diagonal pair"""
    idp = p.variant == abi.VARIANT_PRV_IDP
    nf, nk = p.n_kf_free, p.n_kf
    ob = p.pt_obs_begin
    key = np.empty(p.n_pt, np.int64)
    for i in range(p.n_pt):
        ks = list(p.obs_kf[ob[i]:ob[i + 1]])
        key[i] = min(ks + ([p.pt_ref_kf[i]] if idp else [nk - 1]))
    lm_order = np.lexsort((np.arange(p.n_pt), key))
    rank = np.empty(p.n_pt, np.int64); rank[lm_order] = np.arange(p.n_pt)
    obs_pt = np.repeat(np.arange(p.n_pt), np.diff(ob))
    order = np.lexsort((np.arange(p.n_obs), rank[obs_pt], p.obs_kf))         # by (keyframe, rank of the landmark)
    slot_perm = np.empty(p.n_obs, np.int64); slot_perm[order] = np.arange(p.n_obs)
    kf_seg = np.concatenate([[0], np.cumsum(np.bincount(p.obs_kf, minlength=nk))])
    if idp:
        po = np.lexsort((rank, p.pt_ref_kf))
        pt_perm = np.empty(p.n_pt, np.int64); pt_perm[po] = np.arange(p.n_pt)
        ref_seg = np.concatenate([[0], np.cumsum(np.bincount(p.pt_ref_kf, minlength=nk))])
    else:
        pt_perm, ref_seg = np.arange(p.n_pt), np.zeros(nk + 1, np.int64)
    plain, refit = {}, {}
    for i in range(p.n_pt):
        tr = [(int(p.obs_kf[o]), int(slot_perm[o])) for o in range(ob[i], ob[i + 1])]
        if idp:
            tr.append((int(p.pt_ref_kf[i]), p.n_obs + int(pt_perm[i])))
        tr = sorted(t for t in tr if t[0] < nf)
        for x in range(len(tr)):
            for y in range(x + 1, len(tr)):
                isref = tr[x][1] >= p.n_obs or tr[y][1] >= p.n_obs
                (refit if isref else plain).setdefault((tr[x][0], tr[y][0]), []).append((tr[x][1], tr[y][1]))
    return obs_pt, slot_perm, kf_seg, pt_perm, ref_seg, plain, refit


@pytest.mark.parametrize("variant,kw", [
    (abi.VARIANT_PRV_IDP, dict(n_kf=12, n_fixed=1, n_pt=400, n_obs=2400, seed=41)),
    (abi.VARIANT_PRV_IDP, dict(n_kf=9, n_fixed=3, n_pt=150, n_obs=700, seed=42)),      # fixed reference keyframes, fixed observers
    (abi.VARIANT_PRV_IDP, dict(n_kf=70, n_fixed=1, n_pt=900, n_obs=5400, seed=44)),    # > 64 keyframes: two-word landmark masks
    (abi.VARIANT_SE3_XYZ, dict(n_kf=10, n_fixed=2, n_pt=300, n_obs=1800, seed=43)),
])
def test_device_structure_equals_numpy_reconstruction(ba, variant, kw):
    algo = abi.ALGO_GN if variant == abi.VARIANT_PRV_IDP else abi.ALGO_LM
    p = _shuffled(synth.make_window(variant, algo=algo, **kw), 7, drop_middle=True)
    other = synth.make_window(variant, algo=algo, n_kf=6, n_fixed=1 if variant == abi.VARIANT_PRV_IDP else 2, n_pt=50, n_obs=200, seed=3)
    ba.upload([other, p])      # second window of a batch: every offset is exercised
    nf, nk, npairs = p.n_kf_free, p.n_kf, p.n_kf_free * (p.n_kf_free + 1) // 2
    npairs0 = other.n_kf_free * (other.n_kf_free + 1) // 2
    obs_pt, slot_perm, kf_seg, pt_perm, ref_seg, plain, refit = _expected(p)
    assert (_buf(ba, "OBSPT", np.int32, p.n_obs, 4 * other.n_obs) == obs_pt).all()
    assert (_buf(ba, "SLOTPERM", np.int32, p.n_obs, 4 * other.n_obs) == slot_perm).all()
    assert (_buf(ba, "PTPERM", np.int32, p.n_pt, 4 * other.n_pt) == pt_perm).all()
    assert (_buf(ba, "KFSEG", np.int32, nk + 1, 4 * (other.n_kf + 1)) == kf_seg).all()
    assert (_buf(ba, "REFSEG", np.int32, nk + 1, 4 * (other.n_kf + 1)) == ref_seg).all()
    ib = _buf(ba, "ITEMBEG", np.int32, npairs + 1, 4 * (npairs0 + 1))
    im = _buf(ba, "ITEMMID", np.int32, npairs + 1, 4 * (npairs0 + 1))
    desc_item0 = sum((m + 1) * m // 2 if variant == abi.VARIANT_PRV_IDP else m * (m - 1) // 2 for m in np.diff(other.pt_obs_begin))
    items = _buf(ba, "ITEMS", np.int32, 2 * int(ib[-1]), 8 * int(desc_item0)).reshape(-1, 2)
    assert ib[0] == 0 and (np.diff(ib) >= 0).all() and (im >= ib).all() and (im[:-1] <= ib[1:]).all()
    n_items = 0
    for a in range(nf):
        for b in range(a, nf):
            pi = a * nf - a * (a - 1) // 2 + (b - a)
            got_plain = [tuple(x) for x in items[ib[pi]:im[pi]]]
            got_ref = [tuple(x) for x in items[im[pi]:ib[pi + 1]]]
            if a == b:
                assert not got_plain and not got_ref          # the diagonal pair is two index ranges, not a list
                continue
            exp_plain, exp_ref = plain.get((a, b), []), refit.get((a, b), [])
            # two observation records: in the record order of keyframe a (= lm_order); reference items: as a set
            assert got_plain == sorted(exp_plain), (a, b)
            assert sorted(got_ref) == sorted(exp_ref), (a, b)
            n_items += len(got_plain) + len(got_ref)
    assert n_items == ib[-1] > 0
    # and the solve on this structure is the oracle's (the parity suite proper checks that at every size)
    ba.run()
    _, rs = ba.download()
    assert rs[1].status == 0 and rs[1].its_done[0] >= 1
