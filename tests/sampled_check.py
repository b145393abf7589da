"""Independent check of an operator application at sizes where a host CSR of the whole matrix is out of reach:
a few thousand rows are exported (isph_mat_export_rows), multiplied on the host, and compared with the device SpMV."""
import numpy as np


def assert_spmv_matches_host_on_sampled_rows(A, x, y, nsamples=64, rows_per_sample=64, tol=1e-12):
    """y = A.spmv(x) as the device computed it; x, y torch or numpy.  The exported rows share no code with the SpMV
    kernels (k_sell_rows_to_csr walks the 32-bit column image; the production SpMV reads the 16-bit window columns)."""
    rows, ax = A.sampled_product(x, nsamples=nsamples, rows_per_sample=rows_per_sample)
    yh = y.detach().cpu().numpy() if hasattr(y, "detach") else np.asarray(y)
    scale = max(float(A.last_sample_magnitude.max()), 1e-300)      # size of the terms a row sums, not of the sum
    assert len(rows) >= min(nsamples * rows_per_sample, A.info()["nrow"]) // 2
    assert np.max(np.abs(yh[rows] - ax)) <= tol * scale, (np.max(np.abs(yh[rows] - ax)), scale)
    return rows, ax
