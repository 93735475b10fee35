"""Row-wise parity statistics shared by `__graft_entry__.smoke()` and the GPU parity tests."""


def row_parity(got, ref, tol=1e-5):
    """Per-ROW distance of `got` from `ref` against north_star's "within 1e-5 fp32" written row by row:
    |d_i| <= tol * (1 + |ref_i|).  Returns the worst row's distance in units of its own bound (`worst_scaled`, its
    index and raw numbers), the fraction of rows beyond their bound (`exceed_frac`) and -- for the record -- the
    fraction beyond the ABSOLUTE `tol` (`abs_exceed_frac`: log p of a 10-D flow is a sum of ~25 log-determinants of
    magnitude ~1..10, so one fp32 ulp of the sum is already 1e-6..8e-6 and the eager fp32 reference itself misses an
    absolute 1e-5 on ~10 % of the rows against its own fp64 evaluation)."""
    got, ref = got.detach().double().reshape(-1), ref.detach().double().reshape(-1)
    d = (got - ref).abs()
    bound = tol * (1.0 + ref.abs())
    scaled = d / bound
    i = int(scaled.argmax()) if scaled.numel() else 0
    return {"worst_scaled": float(scaled.max()) if scaled.numel() else 0.0, "worst_row": i,
            "worst_abs": float(d[i]) if scaled.numel() else 0.0, "worst_ref": float(ref[i]) if scaled.numel() else 0.0,
            "exceed_frac": float((scaled > 1.0).double().mean()) if scaled.numel() else 0.0,
            "abs_exceed_frac": float((d > tol).double().mean()) if scaled.numel() else 0.0,
            "max_abs": float(d.max()) if scaled.numel() else 0.0}
