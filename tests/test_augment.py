"""Host augmentation (alphapig_amd.augment) against the reference's get_equi_data tables."""
import os

import numpy as np

from alphapig_amd.augment import dihedral_tables, get_equi_data


def test_tables_and_tuples_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "equi.npz"))
    for k in range(int(g["n"])):
        w = int(g["e%d_w" % k])
        ps, pp = dihedral_tables(w)
        np.testing.assert_array_equal(ps, g["e%d_state_perm" % k][:, 0].reshape(8, -1))
        np.testing.assert_array_equal(pp, g["e%d_pi_perm" % k])
        st = g["e%d_in_state" % k].astype(np.float64)
        pi = g["e%d_in_pi" % k]
        ext = get_equi_data([(st[0], pi[0], 1.0), (st[1], pi[1], -1.0)], w, w)
        assert len(ext) == 16 and ext[0][0].shape == (9, w, w)
        np.testing.assert_array_equal(np.stack([e[0] for e in ext]).astype(np.uint8), g["e%d_out_state" % k])
        np.testing.assert_array_equal(np.stack([e[1] for e in ext]), g["e%d_out_pi" % k])
        np.testing.assert_array_equal(np.array([e[2] for e in ext]), g["e%d_out_z" % k])


def test_group_properties():
    """size-independent: every image is a permutation; the 8 images are distinct on an
    asymmetric board; identity is entry 6."""
    ps, pp = dihedral_tables(15)
    for t in (ps, pp):
        assert all(sorted(r) == list(range(225)) for r in t)
        assert len({tuple(r) for r in t}) == 8
    np.testing.assert_array_equal(ps[6], np.arange(225))
    np.testing.assert_array_equal(pp[6], np.arange(225))
