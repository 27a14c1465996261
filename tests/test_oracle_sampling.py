"""CPU: the temperature-sampling restatement (oracle.qwen25vl.sample_temperature) against HF's own logits warper /
softmax and against its defining properties.  The reference samples with do_sample=True, temperature=0.01,
top_k = top_p = None (src/eval/infer.py:109-115, 160-162)."""
import numpy as np
import pytest

from oracle import qwen25vl as Q


def rand_logits(seed, vocab=2048, scale=3.0):
    return (np.random.default_rng(seed).normal(size=vocab) * scale).astype(np.float32)


def test_uniform_stream_is_a_pure_function_of_seed_slot_index():
    a = [Q.sample_uniform(5, 2, i) for i in range(64)]
    assert a == [Q.sample_uniform(5, 2, i) for i in range(64)]
    assert all(0.0 <= float(u) < 1.0 for u in a)
    assert a != [Q.sample_uniform(5, 3, i) for i in range(64)]      # another chain slot: another stream
    assert a != [Q.sample_uniform(6, 2, i) for i in range(64)]
    us = np.array([Q.sample_uniform(11, 0, i) for i in range(4000)], dtype=np.float64)
    assert abs(us.mean() - 0.5) < 0.02 and abs((us < 0.25).mean() - 0.25) < 0.03


def test_distribution_matches_hf_warper_and_softmax():
    torch = pytest.importorskip("torch")
    lp = pytest.importorskip("transformers.generation.logits_process")
    lg = rand_logits(1)
    seen = [3, 77, 1500, 219]
    for temperature, penalty in ((1.0, 1.0), (0.7, 1.3)):
        scores = torch.from_numpy(lg.copy())[None]
        ids = torch.tensor([seen])
        if penalty != 1.0:
            scores = lp.RepetitionPenaltyLogitsProcessor(penalty)(ids, scores)
        scores = lp.TemperatureLogitsWarper(temperature)(ids, scores)
        probs = torch.softmax(scores, dim=-1)[0].numpy().astype(np.float64)   # what torch.multinomial draws from
        n = 6000
        cnt = np.zeros(lg.shape[0])
        for i in range(n):
            tok, _ = Q.sample_temperature(lg, seen, penalty, temperature, seed=9, slot=1, index=i)
            cnt[tok] += 1
        top = np.argsort(-probs)[:8]
        for k in top:  # binomial 5-sigma band
            sd = np.sqrt(probs[k] * (1 - probs[k]) / n)
            assert abs(cnt[k] / n - probs[k]) < 5 * sd + 1e-3, (temperature, penalty, int(k))
        assert cnt[probs < 1e-7].sum() <= 2  # ~2e-5 of total mass sits there: 0.1 expected hits


def test_low_temperature_is_argmax_unless_tied():
    lg = rand_logits(2)
    assert all(Q.sample_temperature(lg, [], 1.0, 0.01, 3, 0, i)[0] == int(lg.argmax()) for i in range(50))
    lg2 = lg.copy()
    a = int(lg.argmax())
    b = (a + 7) % lg.shape[0]
    lg2[b] = lg2[a]  # exact tie: both must show up, nothing else
    got = {Q.sample_temperature(lg2, [], 1.0, 0.01, 3, 0, i)[0] for i in range(200)}
    assert got == {a, b}


def test_inverse_cdf_rule_and_gap():
    lg = np.full(300, -30.0, dtype=np.float32)
    lg[[5, 150, 299]] = [0.0, np.log(2.0), np.log(1.0)]  # masses 1 : 2 : 1
    for i in range(300):
        u = float(Q.sample_uniform(4, 0, i))
        tok, gap = Q.sample_temperature(lg, [], 1.0, 1.0, 4, 0, i)
        want = 5 if u < 0.25 else (150 if u < 0.75 else 299)
        if gap > 1e-5:
            assert tok == want, (i, u)
