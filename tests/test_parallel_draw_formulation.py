"""The lane-parallel restatement of rng.choice(F, 2N, replace=False) that the 64-lane groups run on the GPU
(dl_reference_models_amd/csrc/mapf_kernels.inl, draw_stage_b, LPE == 64) as plain Python, against NumPy's algorithm
written sequentially (Floyd's sampling with "already chosen -> take j", then _shuffle_int's tail shuffle;
numpy/random/_generator.pyx, Generator.choice; the reference calls it at MA-env:267-282).  Host logic only: the bounded
draws themselves are inputs here.  Adversarial inputs: populations barely larger than the sample (every value in the j
range: long collision chains), few distinct values, swap targets concentrated on a few positions (deep forests)."""
import numpy as np
import pytest


def sequential(vals, J, pop, size):
    base, chosen_set, out = pop - size, set(), []
    for k in range(size):
        v, j = vals[k], base + k
        if v in chosen_set:
            chosen_set.add(j)
            out.append(j)
        else:
            chosen_set.add(v)
            out.append(v)
    idx = list(out)
    for i in range(size - 1, 0, -1):
        idx[i], idx[J[i]] = idx[J[i]], idx[i]
    return out, idx


def lane_parallel(vals, J, pop, size):
    base = pop - size
    dup = [any(vals[s] == vals[k] for s in range(k)) for k in range(size)]  # equality masks by ballots on the GPU
    coll, rounds = list(dup), 0
    while True:  # coll_k = dup_k | (u_k < k & coll[u_k]), u_k = val_k - base: a chain through lower indices
        rounds += 1
        new = [dup[k] or (vals[k] >= base and vals[k] - base < k and coll[vals[k] - base]) for k in range(size)]
        if new == coll:
            break
        coll = new
    chosen = [base + k if coll[k] else vals[k] for k in range(size)]
    Jx = [0] + list(J[1:])
    nx = [min([t for t in range(max(i + 1, 1), size) if Jx[t] == Jx[i]], default=None) for i in range(size)]
    ptr = [min([t for t in range(max(x + 1, 1), size) if Jx[t] == x], default=x) for x in range(size)]  # up(x) or a root
    jumps = 0
    while True:  # pointer doubling to the roots
        jumps += 1
        nxt = [ptr[ptr[x]] for x in range(size)]
        if nxt == ptr:
            break
        ptr = nxt
    final = [chosen[ptr[nx[i]]] if nx[i] is not None else chosen[Jx[i]] for i in range(size)]
    return chosen, final, rounds, jumps


@pytest.mark.parametrize("mode", range(5))
def test_lane_parallel_choice_equals_numpys_sequential_algorithm(mode):
    rng = np.random.default_rng(100 + mode)
    worst_jumps = 0
    for trial in range(1500):
        N = int(rng.integers(2, 65))
        size = 2 * N
        pop = int(rng.integers(size + 1, size + 1 + (5 if trial % 3 == 0 else 4000)))
        base = pop - size
        vals = [int(rng.integers(0, base + k + 1)) for k in range(size)]
        if mode == 1:
            vals = [int(rng.integers(max(0, base - 2), base + k + 1)) for k in range(size)]
        if mode == 2:
            vals = [min(base + k, int(rng.integers(0, 3)) + (base if k % 2 else 0)) for k in range(size)]
        J = [0] + [int(rng.integers(0, i + 1)) for i in range(1, size)]
        if mode == 3:
            J = [0] + [int(rng.integers(0, min(i, 2) + 1)) for i in range(1, size)]
        if mode == 4:
            J = [0] + [max(0, i - int(rng.integers(0, 2))) for i in range(1, size)]
        want_chosen, want_idx = sequential(vals, J, pop, size)
        chosen, final, _, jumps = lane_parallel(vals, J, pop, size)
        assert chosen == want_chosen, (trial, vals)
        assert final == want_idx, (trial, J)
        worst_jumps = max(worst_jumps, jumps)
    assert worst_jumps <= 8  # the kernel's pointer-doubling loop runs at most 8 rounds (2^7 >= 128)
