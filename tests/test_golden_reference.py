"""Pins against the REAL reference: tests/golden/reference_intree.pt was produced by
tools/make_golden.py importing /root/reference's own functions (the parts of the hot
path that live in the reference tree).  Both the product helpers (sbi_amd.utils...) and
the oracle's restatements must reproduce them."""

import os

import pytest
import torch

G = torch.load(os.path.join(os.path.dirname(__file__), "golden", "reference_intree.pt"), weights_only=False)


def test_searchsorted_reference_vectors():
    """Vectors of tests/torchutils_test.py:138-158 evaluated by the real reference: left edges and
    mid-points of the 9 bins -> arange(9); an interior right edge belongs to the NEXT bin and only the
    last right edge (knot + 1e-6) stays in the last bin."""
    from oracle.nsf_oracle import searchsorted as o_ss
    from sbi_amd.utils.torchutils import searchsorted as p_ss

    d = G["searchsorted"]
    for ss in (o_ss, p_ss):
        for key in ("left", "right", "mid"):
            got = ss(d["bins"][None, :].clone(), d[key])
            assert torch.equal(got, d["idx_" + key])
        assert torch.equal(d["idx_left"], torch.arange(9)) and d["idx_right"].tolist() == [1, 2, 3, 4, 5, 6, 7, 8, 8]
        r = G["searchsorted_rand"]
        assert torch.equal(ss(r["knots"].clone(), r["x"]), r["idx"])


def test_masks_repeat_rows_sum_except_batch():
    from oracle import nsf_oracle as O
    from sbi_amd.utils import torchutils as P

    for key, ref in G["masks"].items():
        d, e = key.split("_")
        for mod in (O, P):
            assert torch.equal(mod.create_alternating_binary_mask(int(d), even=bool(int(e))), ref)
    rr = G["repeat_rows"]
    assert torch.equal(O.repeat_rows(rr["x"], 3), rr["out"]) and torch.equal(P.repeat_rows(rr["x"], 3), rr["out"])
    sb = G["sum_except_batch"]
    assert torch.allclose(O.sum_except_batch(sb["x"]), sb["out"])


def test_zscore_statistics_match_reference():
    from oracle.nsf_oracle import z_standardization as o_z
    from sbi_amd.utils.sbiutils import standardizing_stats, z_standardization

    batch, stats = G["zscore"]["batch"], G["zscore"]["stats"]
    for structured in (False, True):
        m, s = z_standardization(batch, structured)
        assert torch.equal(m, stats[f"theta_{structured}"]["mean"]) and torch.equal(s, stats[f"theta_{structured}"]["std"])
        m, s = o_z(batch, structured, 1e-14)
        assert torch.equal(m, stats[f"theta_{structured}"]["mean"]) and torch.equal(s, stats[f"theta_{structured}"]["std"])
        m, s = standardizing_stats(batch, structured)
        assert torch.equal(m, stats[f"x_{structured}"]["mean"]) and torch.equal(s, stats[f"x_{structured}"]["std"])
    m, s = standardizing_stats(batch[:1])
    assert torch.equal(s, stats["x_single_row"]["std"])


def test_zscore_parser_and_invalid_x():
    from sbi_amd.utils.sbiutils import handle_invalid_x, z_score_parser

    for k, ref in G["z_score_parser"].items():
        assert z_score_parser(None if k == "None" else k) == ref
    h = G["handle_invalid_x"]
    valid, nn_, ni = handle_invalid_x(h["x"], True)
    assert torch.equal(valid, h["out"][0]) and (nn_, ni) == (h["out"][1], h["out"][2])


def test_shape_handling_matches_reference():
    from sbi_amd.neural_nets.estimators.shape_handling import reshape_to_sample_batch_event

    for name, d in G["shape_handling"].items():
        out = reshape_to_sample_batch_event(d["inp"], torch.Size((4,)), leading_is_sample=d["lead"])
        assert out.shape == d["out"].shape and torch.equal(out, d["out"]), name


def test_within_support_and_box_uniform():
    from sbi_amd.utils.sbiutils import within_support
    from sbi_amd.utils.torchutils import BoxUniform

    d = G["within_support"]
    box = BoxUniform(-2 * torch.ones(3), 2 * torch.ones(3))
    assert torch.equal(within_support(box, d["pts"]), d["inside"])
    assert torch.equal(torch.isfinite(box.log_prob(d["pts"])), torch.isfinite(d["logp"]))


def test_linear_gaussian_simulator_and_true_posterior():
    from sbi_amd.simulators.linear_gaussian import linear_gaussian, true_posterior_linear_gaussian_mvn_prior

    lg = G["linear_gaussian"]
    for dim in (2, 10):
        shift, cov = -1.0 * torch.ones(dim), 0.3 * torch.eye(dim)
        post = true_posterior_linear_gaussian_mvn_prior(torch.zeros(1, dim), shift, cov, torch.zeros(dim),
                                                        torch.eye(dim))
        assert torch.allclose(post.mean, lg[dim]["mean"], atol=1e-7)
        assert torch.allclose(post.covariance_matrix, lg[dim]["cov"], atol=1e-7)
        torch.manual_seed(7)
        theta = torch.randn(16, dim)
        assert torch.equal(theta, lg[dim]["theta"])
        assert torch.allclose(linear_gaussian(theta, shift, cov), lg[dim]["sim"], atol=1e-6)
    post = true_posterior_linear_gaussian_mvn_prior(torch.full((1, 10), 0.25), torch.zeros(10), 0.1 * torch.eye(10),
                                                    torch.zeros(10), 0.1 * torch.eye(10))
    assert torch.allclose(post.mean, lg["mini_sbibm"]["mean"], atol=1e-7)


def test_accept_reject_sample_reproduces_reference_run():
    """Same seed, same toy proposal: the device-compaction sampler returns the reference's samples."""
    from sbi_amd.samplers.rejection.rejection import accept_reject_sample
    from sbi_amd.utils.sbiutils import within_support
    from sbi_amd.utils.torchutils import BoxUniform

    box = BoxUniform(-2 * torch.ones(3), 2 * torch.ones(3))

    def proposal(shape, condition):
        return torch.randn(shape[0], condition.shape[0], 3) * 0.8

    torch.manual_seed(3)
    smp, acc = accept_reject_sample(proposal, lambda t: within_support(box, t), 5000, max_sampling_batch_size=700,
                                    proposal_sampling_kwargs={"condition": torch.zeros(2, 5)})
    assert torch.equal(smp, G["accept_reject"]["samples"])
    assert torch.allclose(acc, G["accept_reject"]["acceptance"])


def test_accept_reject_all_accepted_first_pass_returns_the_candidates_in_order():
    """The fast path of `accept_reject_sample` (a request-sized first pass in which every candidate is accepted) must
    return exactly what the compaction would have produced: the candidates, in the order they were drawn, acceptance 1;
    one rejected candidate sends the call down the general path with the same result as before."""
    from sbi_amd.samplers.rejection.rejection import accept_reject_sample

    def proposal(shape, condition=None):
        return torch.randn(shape[0], 1, 3)

    torch.manual_seed(4)
    ref = proposal(torch.Size((500,)))
    torch.manual_seed(4)
    s, acc = accept_reject_sample(proposal, lambda th: torch.ones(th.shape[:2], dtype=torch.bool), 500,
                                  max_sampling_batch_size=500)
    assert torch.equal(s, ref) and torch.equal(acc, torch.ones(1))
    # not request-sized (two passes of 250): general path, same samples
    torch.manual_seed(4)
    s2, acc2 = accept_reject_sample(proposal, lambda th: torch.ones(th.shape[:2], dtype=torch.bool), 500,
                                    max_sampling_batch_size=250)
    torch.manual_seed(4)
    ref2 = torch.cat([proposal(torch.Size((250,))), proposal(torch.Size((250,)))])
    assert torch.equal(s2, ref2) and torch.allclose(acc2, torch.ones(1))
    # one rejection in a request-sized pass: the general path compacts and tops up
    torch.manual_seed(4)
    s3, acc3 = accept_reject_sample(proposal, lambda th: th[..., 0] > -1.0, 500, max_sampling_batch_size=500)
    assert s3.shape == (500, 1, 3) and bool((s3[..., 0] > -1.0).all()) and float(acc3) < 1.0
