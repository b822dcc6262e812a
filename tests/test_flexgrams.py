"""Flexgrams abstracted from skipgrams: IndexedPatternModel::computeflexgrams_fromskipgrams
(reference include/patternmodel.h:3724-3744).

CPU: the restatement in oracle/oracle.py against the reference's known answer (src/test.cpp:1441-1443: 22 flexgrams, model size
155) and against dumps of the reference itself (tests/golden/flex.*.txt, written by make_golden.py where the reference's
insert-while-iterating loop stayed stable). GPU: colibri_flexgrams / colibri_flexgrams_fetch through the C ABI against the
restatement — identical flexgram set, counts and reference lists.
"""
import glob
import os

import numpy as np
import pytest

from conftest import small_corpora

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _flex_only(model):
    import oracle
    return {k: v for k, v in model.refs.items() if oracle.key_category(k) == 3}


def test_known_answer_hamlet_22_flexgrams():
    """src/test.cpp:1441-1443: the indexed skipgram model of hamlet (133 patterns) yields 22 flexgrams, 155 patterns in all."""
    import oracle
    m = oracle.parse_dump(open(os.path.join(GOLD, "hamlet.v2.is.l100.txt")).read(), indexed=True)
    assert len(m) == 133
    f, found = oracle.flexgrams_from_skipgrams(m)
    assert found == 22 and len(f) == 155


def test_toflexgram_and_category():
    import oracle
    assert oracle.toflexgram(bytes([6, 3, 3, 7, 3, 8])) == bytes([6, 4, 7, 4, 8])  # "To {*} {*} or {*} to" -> "To {**} or {**} to" (test.py:147-151)
    assert oracle.toflexgram(bytes([0x83, 0x03, 3, 9])) == bytes([0x83, 0x03, 4, 9])  # 03 as the low byte of a 2-byte token is no gap
    assert oracle.key_category(bytes([6, 7])) == 1 and oracle.key_category(bytes([6, 3, 7])) == 2 and oracle.key_category(bytes([6, 4, 7])) == 3
    assert oracle.key_category(bytes([6, 3, 7, 4, 8])) == 3


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "flex.*.txt"))), ids=os.path.basename)
def test_restatement_matches_reference_dumps(path):
    """flex.<corpus>.<mode>.txt = the reference's model after computeflexgrams_fromskipgrams; <corpus>.<mode>.l5.txt = before."""
    import oracle
    name = os.path.basename(path)[len("flex."):-len(".txt")]
    before = oracle.parse_dump(open(os.path.join(GOLD, name + ".l5.txt")).read(), indexed=True)
    after = oracle.parse_dump(open(path).read(), indexed=True)
    mine, found = oracle.flexgrams_from_skipgrams(before)
    assert mine.counts == after.counts and mine.refs == after.refs
    assert found == len(after) - len(before) > 0


# ------------------------------------------------------------------------------------------------------------------------------
def _arrays_of(model):
    keys = list(model.refs)
    key_off = np.zeros(len(keys) + 1, dtype=np.uint64)
    key_off[1:] = np.cumsum([len(k) for k in keys])
    ref_off = np.zeros(len(keys) + 1, dtype=np.uint64)
    ref_off[1:] = np.cumsum([len(model.refs[k]) for k in keys])
    kb = np.frombuffer(b"".join(keys) or b"\0", dtype=np.uint8)
    rs = np.array([r[0] for k in keys for r in model.refs[k]], dtype=np.uint32)
    rt = np.array([r[1] for k in keys for r in model.refs[k]], dtype=np.uint16)
    return key_off, kb, ref_off, rs, rt


def _device_flexgrams(ctx, model):
    fo, fk, fc, (fro, frs, frt) = ctx.flexgrams(*_arrays_of(model))
    kb, off, ro = fk.tobytes(), fo.tolist(), fro.tolist()
    rs, rt = frs.tolist(), frt.tolist()
    out = {}
    for j, c in enumerate(fc.tolist()):
        k = kb[off[j]: off[j + 1]]
        assert k not in out
        out[k] = list(zip(rs[ro[j]: ro[j + 1]], rt[ro[j]: ro[j + 1]]))
        assert c == len(out[k])
    return out


@pytest.fixture(scope="module")
def ctx():
    from colibri_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hamlet.v2.is.l100", "hamlet.v2.is.l5", "hamlet.v2.isT1.l5", "phrases15k.is.l5", "phrases15k.isT1.l5", "zipf20k.is.l5"])
def test_device_flexgrams_on_reference_models(ctx, name):
    import oracle
    m = oracle.parse_dump(open(os.path.join(GOLD, name + ".txt")).read(), indexed=True)
    want, found = oracle.flexgrams_from_skipgrams(m)
    got = _device_flexgrams(ctx, m)
    assert got == _flex_only(want) and len(got) == found


@pytest.mark.gpu
@pytest.mark.parametrize("corpus", ["rand1", "rand2", "zipf20k", "zipf200k_phrases"])
@pytest.mark.parametrize("maxlength,minskiptypes", [(5, 1), (7, 2)])
def test_device_flexgrams_after_device_training(ctx, corpus, maxlength, minskiptypes):
    """train on the device (indexed + skipgrams), export, abstract on the device; the oracle does both steps on the CPU"""
    import oracle
    payload = small_corpora()[corpus]
    ctx.upload(payload)
    ctx.train(mintokens=2, maxlength=maxlength, indexed=1, doskipgrams=1, minskiptypes=minskiptypes)
    key_off, key_bytes, counts, (ref_off, rs, rt) = ctx.export_arrays()
    fo, fk, fc, (fro, frs, frt) = ctx.flexgrams(key_off, key_bytes, ref_off, rs, rt)
    om = oracle.train(payload, 2, maxlength, indexed=True, doskipgrams=True, minskiptypes=minskiptypes)
    want, found = oracle.flexgrams_from_skipgrams(om)
    kb, off, ro = fk.tobytes(), fo.tolist(), fro.tolist()
    got = {kb[off[j]: off[j + 1]]: list(zip(frs[ro[j]: ro[j + 1]].tolist(), frt[ro[j]: ro[j + 1]].tolist())) for j in range(len(fc))}
    assert got == _flex_only(want) and len(got) == found
    assert fc.tolist() == [len(got[kb[off[j]: off[j + 1]]]) for j in range(len(fc))]
    # the same without the host round trip: on the model still resident in HBM
    ro, rk, rc, (rro, rrs, rrt) = ctx.flexgrams_resident()
    assert (ro.tolist(), rk.tobytes(), rc.tolist(), rro.tolist(), rrs.tolist(), rrt.tolist()) == (fo.tolist(), fk.tobytes(), fc.tolist(), fro.tolist(), frs.tolist(), frt.tolist())


@pytest.mark.gpu
def test_resident_flexgrams_need_an_indexed_model(ctx):
    from colibri_amd import capi
    ctx.upload(small_corpora()["rand1"])
    ctx.train(mintokens=2, maxlength=5)
    with pytest.raises(capi.ColibriError):
        ctx.flexgrams_resident()
    ctx.train(mintokens=2, maxlength=5, indexed=1)  # indexed, no skipgrams: nothing to abstract
    fo, fk, fc, _ = ctx.flexgrams_resident()
    assert fc.size == 0 and fo.tolist() == [0]


@pytest.mark.gpu
def test_device_flexgrams_edge_inputs(ctx):
    import oracle
    # no patterns at all / no skipgrams / a pattern that is already a flexgram is not a skipgram / duplicates are kept / unsorted input
    empty = oracle.Model(0, 0, {}, {})
    assert _device_flexgrams(ctx, empty) == {}
    plain = oracle.Model(0, 0, {bytes([6]): 1, bytes([6, 7]): 1}, {bytes([6]): [(1, 0)], bytes([6, 7]): [(1, 0)]})
    assert _device_flexgrams(ctx, plain) == {}
    refs = {
        bytes([6, 3, 7]): [(3, 1), (1, 0)],                    # unsorted on purpose
        bytes([6, 3, 3, 7]): [(1, 0), (2, 5)],                  # same flexgram, one duplicate reference
        bytes([6, 3, 7, 4, 8]): [(9, 9)],                       # has a {**}: category flexgram, ignored
        bytes([0x83, 0x03, 3, 9]): [(70000, 65535)],            # 2-byte token whose low byte is 03
        bytes([6, 3, 7, 3, 3, 3, 8]): [(5, 5)],
        bytes([6, 7, 8]): [(1, 0), (4, 4)],
        bytes([9, 3, 9]): [],                                   # a skipgram without references still names its flexgram
    }
    m = oracle.Model(0, 0, {k: len(v) for k, v in refs.items()}, refs)
    want, found = oracle.flexgrams_from_skipgrams(m)
    got = _device_flexgrams(ctx, m)
    assert got == {k: v for k, v in _flex_only(want).items() if k != bytes([6, 3, 7, 4, 8])}
    assert got[bytes([6, 4, 7])] == [(1, 0), (1, 0), (2, 5), (3, 1)]
    assert got[bytes([9, 4, 9])] == []
