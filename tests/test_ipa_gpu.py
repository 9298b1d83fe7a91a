"""GPU parity for the InnerProductArgPC halving rounds (ipa_pc/mod.rs:664-711) against the
oracle's restatement, with the round challenges supplied (the Fiat-Shamir hash stays on the
host with the caller)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve,n", [("pallas", 1 << 10), ("pallas", 4), ("bls12_381", 1 << 8), ("bn254", 1 << 9)])
def test_ipa_open_rounds(ctx, curve, n):
    import torch
    from poly_commit_amd import ipa
    lg = n.bit_length() - 1
    key = O.gen_bases(curve, n + 1)
    comm_key, h_prime = key[:n], key[n]
    coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xA11CE, n))
    point = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xB0B, 1))[0]
    ch = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xC4A1, lg))
    want_l, want_r, want_key, want_c = O.ipa_rounds(curve, np.ascontiguousarray(comm_key), coeffs, point,
                                                    np.ascontiguousarray(h_prime), ch)
    it = iter(range(lg))
    cdev = torch.from_numpy(coeffs.view(np.int64).copy()).cuda()
    l, r, fk, c = ipa.ipa_open_rounds(ctx, curve, comm_key, cdev, n, point, h_prime, lambda L, R_: ch[next(it)])
    assert (l == want_l).all() and (r == want_r).all()
    assert (fk == want_key).all() and (c == want_c).all()
