"""The oracle's closed-form solves against the compiled reference Matrix class
(golden vectors from oracle/_ref; live comparison too when it is built)."""
import os

import numpy as np
import pytest

import oracle_lib as O


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "solve_golden.npz"))


def test_p2p_rotation_matches_reference_svd(G):
    # icpPointToPoint.cpp:160-162 via matrix.cpp:582 svd
    for H, R_ in zip(G["H"], G["R_"]):
        assert np.abs(O.p2p_rotation(H).reshape(4) - R_).max() < 1e-12


def test_p2p_rotation_reflection_branch(G):
    dets = G["H"][:, 0] * G["H"][:, 3] - G["H"][:, 1] * G["H"][:, 2]
    assert (dets < 0).sum() > 50  # fixture really holds reflections
    for H in G["H"][dets < 0][:50]:
        R_ = O.p2p_rotation(H)
        assert np.linalg.det(R_) == pytest.approx(-1.0, abs=1e-12)


def test_fitstep_solve_half_matches_reference(G):
    # icpPointToPoint.cpp:149-171 replayed with the reference Matrix ops
    model = O.IcpModel(np.random.RandomState(0).randn(8, 2), np.zeros((0, 2)))
    assert model.valid
    for k in range(len(G["fs_n"])):
        n = int(G["fs_n"][k])
        pm, pt = G["fs_pm"][k][:n], G["fs_pt"][k][:n]
        R, t = G["fs_R"][k].reshape(2, 2), G["fs_t"][k]
        # same arithmetic in numpy, following the oracle's statement order
        mu_m, mu_t = pm.sum(0) / n, pt.sum(0) / n
        H = (pt - mu_t).T @ (pm - mu_m)
        R_ = O.p2p_rotation(H)
        t_ = mu_m - R_ @ mu_t
        Rn, tn = R_ @ R, R_ @ t + t_
        d = max(np.linalg.norm(R_ - np.eye(2)), np.linalg.norm(t_))
        assert np.abs(Rn.reshape(4) - G["fs_Ro"][k]).max() < 1e-11
        assert np.abs(tn - G["fs_to"][k]).max() < 1e-10
        assert abs(d - G["fs_d"][k]) < 1e-10


def test_solve3_bit_exact(G):
    # matrix.cpp:420-508; same operation order => identical doubles
    for A, b, x, ok in zip(G["s3_A"], G["s3_b"], G["s3_x"], G["s3_ok"]):
        ok2, x2 = O.solve3(A, b)
        assert ok2 == ok
        if ok:
            assert np.array_equal(x2, x)


def test_orthonormal_from_omega(G):
    # icpPointToPlane.cpp:88-95 (on inputs where the reference svd converged)
    assert len(G["om_w"]) > 250
    for w, R_ in zip(G["om_w"], G["om_R"]):
        assert np.abs(O.orthonormal_from_omega(w).reshape(4) - R_).max() < 1e-12


def test_normal_matches_reference_svd_column(G):
    # icpPointToPlane.cpp:279-305: normal = column 1 of U from H.svd on the neighbours' scatter; the oracle's closed form
    # is that vector up to the svd's sign (which the step does not see: A and b flip together)
    assert len(G["nm_k"]) == 600
    for P, k, n_ref in zip(G["nm_P"], G["nm_k"], G["nm_n"]):
        n = O.normal2(P[:k])
        sgn = 1.0 if n @ n_ref >= 0 else -1.0
        assert np.abs(n - sgn * n_ref).max() < 1e-12
        assert abs(n @ n - 1.0) < 1e-14


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built")
def test_live_reference_normals_of_a_model():
    # the oracle model's own neighbour sets (K = 10) through the compiled reference
    from slam_amd import synth
    m_ga, m_nga = synth.make_map(n_points=2000)
    model = O.IcpModel(m_ga, m_nga, normals_k=10)
    nrm = model.normals()
    pts = np.concatenate([m_ga, m_nga]).astype(np.float32)
    d2 = ((pts[:, None, :].astype(np.float64) - pts[None, :, :].astype(np.float64)) ** 2).sum(-1)
    for i in range(0, len(pts), 7):
        nb = np.argsort(d2[i], kind="stable")[:10]
        n_ref = O.ref_normal2(pts[nb])
        sgn = 1.0 if nrm[i] @ n_ref >= 0 else -1.0
        assert np.abs(nrm[i] - sgn * n_ref).max() < 1e-9, i


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built")
def test_live_reference_matrix_agrees():
    rs = np.random.RandomState(5)
    for _ in range(2000):
        q = rs.randn(rs.randint(3, 30), 2) * rs.uniform(0.01, 20, 2)
        qm = q + rs.randn(*q.shape) * 1e-2
        H = (q - q.mean(0)).T @ (qm - qm.mean(0))
        assert np.abs(O.p2p_rotation(H) - O.ref_p2p_rotation(H)).max() < 1e-12
    assert np.array_equal(O.p2p_rotation(np.zeros(4)), O.ref_p2p_rotation(np.zeros(4)))
