"""Sparse tile kernels (spartan_amd/csrc/sparse.hip) through the C-ABI against scipy.sparse -- the library
the reference's sparse tile bodies are written in (spartan/array/sparse.pyx, tile.pyx:226-252, dot.py:212-240).
Structure (indptr / indices) must be identical to scipy's canonical CSR; values are bit-exact wherever no
floating-point re-association is involved (no duplicates / integer-valued data), otherwise within the stated
tolerance."""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu

from spartan_amd import devarray as D  # noqa: E402
from spartan_amd import sparse as S  # noqa: E402

DEV = 'hip'
dev = D.from_numpy


def _rand_coo(rng, m, n, nnz, dtype, dup=False, integer=False):
  rows = rng.randint(0, m, size=nnz).astype(np.int32)
  cols = rng.randint(0, n, size=nnz).astype(np.int32)
  if dup and nnz > 4:
    rows[nnz // 2:] = rows[:nnz - nnz // 2]
    cols[nnz // 2:] = cols[:nnz - nnz // 2]
  vals = rng.randint(-4, 5, size=nnz).astype(dtype) if integer else rng.standard_normal(nnz).astype(dtype)
  return sps.coo_matrix((vals, (rows, cols)), shape=(m, n))


def _canon(mat):
  c = mat.tocsr().copy()
  c.sum_duplicates()
  c.sort_indices()
  return c


def _same_structure(t, ref):
  got = S.to_scipy(t)
  assert got.shape == ref.shape
  np.testing.assert_array_equal(got.indptr, ref.indptr)
  np.testing.assert_array_equal(got.indices, ref.indices)
  return got


@pytest.mark.parametrize('shape,nnz', [((1, 1), 1), ((7, 5), 0), ((13, 1), 9), ((1, 17), 9), ((300, 200), 5000),
                                      ((5000, 70000), 200000), ((2 ** 16, 2 ** 16), 300000)])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_upload_canonical_csr(shape, nnz, dtype):
  rng = np.random.RandomState(nnz + shape[0])
  coo = _rand_coo(rng, shape[0], shape[1], nnz, dtype)
  ref = _canon(coo)
  got = _same_structure(S.from_scipy(coo, DEV), ref)
  # random coordinates collide: colliding values are added in list order, scipy adds them in its own order
  np.testing.assert_allclose(got.data, ref.data, rtol=1e-5 if dtype == np.float32 else 1e-13, atol=1e-6)


def test_upload_duplicates_are_summed_exactly_for_integer_values():
  rng = np.random.RandomState(3)
  coo = _rand_coo(rng, 200, 300, 20000, np.float32, dup=True, integer=True)
  ref = _canon(coo)
  got = _same_structure(S.from_scipy(coo, DEV), ref)
  np.testing.assert_array_equal(got.data, ref.data)


def test_upload_keeps_explicit_zeros_and_any_input_format():
  m = sps.lil_matrix((6, 9), dtype=np.float32)
  m[0, 8] = 2
  m[5, 0] = 0.5
  m[3, 3] = -1
  for fmt in ('lil', 'csr', 'csc', 'coo', 'dok'):
    got = _same_structure(S.from_scipy(m.asformat(fmt), DEV), _canon(m))
    np.testing.assert_array_equal(got.data, _canon(m).data)
  z = sps.coo_matrix((np.array([0.0, 1.0], np.float32), ([1, 2], [1, 2])), shape=(4, 4))
  assert S.from_scipy(z, DEV).nnz == 2     # scipy keeps an explicitly stored zero, so do we


def test_transpose_slice_add_sub():
  rng = np.random.RandomState(5)
  a = _rand_coo(rng, 257, 1031, 9000, np.float32, integer=True)
  b = _rand_coo(rng, 257, 1031, 7000, np.float32, integer=True)
  A, B = S.from_scipy(a, DEV), S.from_scipy(b, DEV)
  ca, cb = _canon(a), _canon(b)
  got = _same_structure(S.transpose(A), _canon(ca.T))
  np.testing.assert_array_equal(got.data, _canon(ca.T).data)
  for (r0, r1, c0, c1) in [(0, 257, 0, 1031), (10, 200, 0, 1031), (0, 257, 100, 101), (250, 257, 1000, 1031),
                           (5, 5, 0, 10)]:
    ref = _canon(ca[r0:r1, c0:c1])
    got = _same_structure(S.slice_box(A, r0, r1, c0, c1), ref)
    np.testing.assert_array_equal(got.data, ref.data)
  # a + b / a - b: scipy drops nothing either (cancelled cells stay stored as 0 only in ours -> compare dense)
  for sign, ref in ((1, ca + cb), (-1, ca - cb)):
    got = S.to_scipy(S.add(A, B, sign))
    np.testing.assert_array_equal(got.toarray(), ref.toarray())
    assert got.has_canonical_format


def test_region_update_matches_compute_sparse_update():
  rng = np.random.RandomState(6)
  old = _canon(_rand_coo(rng, 64, 96, 1500, np.float32, integer=True))
  upd = _canon(_rand_coo(rng, 16, 32, 200, np.float32, integer=True))
  O, U = S.from_scipy(old, DEV), S.from_scipy(upd, DEV)
  box = (8, 24, 40, 72)
  ref = old.tolil()
  ref[box[0]:box[1], box[2]:box[3]] = upd.toarray()
  got = S.to_scipy(S.update_box(O, *box, U, add_to_old=False))
  np.testing.assert_array_equal(got.toarray(), ref.toarray())
  ref2 = old.toarray()
  ref2[box[0]:box[1], box[2]:box[3]] += upd.toarray()
  got2 = S.to_scipy(S.update_box(O, *box, U, add_to_old=True))
  np.testing.assert_array_equal(got2.toarray(), ref2)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('n', [1, 3, 64, 129, 300])
@pytest.mark.parametrize('shape,nnz', [((1000, 777), 20000), ((33, 5000), 40000), ((4096, 64), 900)])
def test_csr_times_dense(shape, nnz, n, dtype):
  rng = np.random.RandomState(n + nnz)
  a = _canon(_rand_coo(rng, shape[0], shape[1], nnz, dtype))
  A = S.from_scipy(a, DEV)
  b = rng.standard_normal((shape[1], n)).astype(dtype)
  got = S.spmm(A, dev(b)).numpy()
  ref = a.astype(np.float64) @ b.astype(np.float64)
  scale = (abs(a).astype(np.float64) @ np.abs(b).astype(np.float64))
  eps = np.finfo(dtype).eps
  k = max(1, int(a.getnnz(axis=1).max()))
  assert np.all(np.abs(got - ref) <= 2 * k * eps * scale + 1e-30)   # |dy| <= nnz_row * eps * sum|a||x|
  if n > 1:
    # n > 1 walks a row's entries in storage order like scipy's csr_matvecs: bit-exact
    np.testing.assert_array_equal(got, a @ b)
  # vectors ([k]) give vectors ([m])
  v = dev(b[:, 0].copy())
  assert tuple(S.spmm(A, v).shape) == (shape[0],)


def test_spmv_integer_valued_is_exact_for_every_group_width(monkeypatch):
  rng = np.random.RandomState(9)
  a = _canon(_rand_coo(rng, 3000, 2000, 60000, np.float32, integer=True))
  A = S.from_scipy(a, DEV)
  x = rng.randint(-3, 4, size=(2000, 1)).astype(np.float32)
  ref = a @ x
  monkeypatch.setenv('SP_SPMV_ALGO', 'vector')
  for g in (2, 4, 8, 16, 32, 64):
    monkeypatch.setenv('SP_SPMV_G', str(g))
    np.testing.assert_array_equal(S.spmm(A, dev(x)).numpy(), ref)
  monkeypatch.delenv('SP_SPMV_G')
  monkeypatch.delenv('SP_SPMV_ALGO')
  np.testing.assert_array_equal(S.row_sums(A).numpy(), np.asarray(a.sum(axis=1)).ravel())
  # accumulate
  y = D.full((3000, 1), 1, np.float32)
  S.spmm(A, dev(x), out=y, accumulate=True)
  np.testing.assert_array_equal(y.numpy(), ref + 1)


@pytest.mark.parametrize('algo', ['stream', 'vector'])
@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_spmv_rows_spanning_entry_chunks(monkeypatch, algo, seed):
  """The stream kernel splits the STORED ENTRIES in chunks of 2048: rows that span several chunks, rows that end
  exactly on a chunk boundary, empty rows at the start / end / on a boundary.  Integer values: exact."""
  monkeypatch.setenv('SP_SPMV_ALGO', algo)
  rng = np.random.RandomState(seed)
  ncols = 9000
  pool = [0, 0, 0, 1, 2, 3, 7, 40, 300, 2048, 2047, 2049, 5000, 4096, 8999]
  lens = [0, 0] + [pool[i] for i in rng.randint(0, len(pool), size=40)] + [0, 0, 0]
  if seed == 0:
    lens = [2048, 0, 0, 2048, 4096, 0, 1, 2047, 0]      # boundaries hit exactly
  rows, cols = [], []
  for r, n in enumerate(lens):
    rows.append(np.full(n, r, np.int32))
    cols.append(np.sort(rng.choice(ncols, size=n, replace=False)).astype(np.int32))
  rows, cols = np.concatenate(rows), np.concatenate(cols)
  vals = rng.randint(-3, 4, size=rows.size).astype(np.float32)
  a = sps.csr_matrix((vals, (rows, cols)), shape=(len(lens), ncols))
  A = S.from_scipy(a, DEV)
  x = rng.randint(-2, 3, size=(ncols, 1)).astype(np.float32)
  np.testing.assert_array_equal(S.spmm(A, dev(x)).numpy(), a @ x)
  # without the per-matrix plan every workgroup searches its own row range
  np.testing.assert_array_equal(S.spmm(A, dev(x), plan=False).numpy(), a @ x)
  np.testing.assert_array_equal(S.row_sums(A).numpy(), np.asarray(a.sum(axis=1)).ravel())
  y = D.full((len(lens), 1), 2.0, np.float32)
  S.spmm(A, dev(x), out=y, accumulate=True)
  np.testing.assert_array_equal(y.numpy(), a @ x + 2)
  # float64 too
  a64 = a.astype(np.float64)
  np.testing.assert_array_equal(S.spmm(S.from_scipy(a64, DEV), dev(x.astype(np.float64))).numpy(),
                                a64 @ x.astype(np.float64))


def test_scatter_modes():
  rng = np.random.RandomState(10)
  a = _canon(_rand_coo(rng, 40, 50, 300, np.float32, integer=True))
  A = S.from_scipy(a, DEV)
  np.testing.assert_array_equal(S.to_dense(A).numpy(), a.toarray())
  base = rng.randint(0, 5, size=(64, 80)).astype(np.float32)
  out = dev(base.copy())
  S.scatter(A, out, 7, 11, mode=1)
  ref = base.copy()
  ref[7:47, 11:61] += a.toarray()
  np.testing.assert_array_equal(out.numpy(), ref)
  # sparse.pyx:21-38 with REDUCE_ADD: first write where the mask is clear, add where it is set
  mask = (rng.rand(64, 80) < 0.5)
  out = dev(base.copy())
  mk = dev(mask.astype(np.uint8))
  S.scatter(A, out, 7, 11, mode=2, mask=mk)
  ref, rmask = base.copy(), mask.copy()
  coo = a.tocoo()
  for r, c, v in zip(coo.row + 7, coo.col + 11, coo.data):
    if rmask[r, c]:
      ref[r, c] += v
    else:
      ref[r, c] = v
      rmask[r, c] = True
  np.testing.assert_array_equal(out.numpy(), ref)
  np.testing.assert_array_equal(mk.numpy().astype(bool), rmask)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_sparse_times_sparse(dtype):
  rng = np.random.RandomState(12)
  a = _canon(_rand_coo(rng, 300, 500, 4000, dtype))
  b = _canon(_rand_coo(rng, 500, 400, 5000, dtype))
  ref = _canon(a @ b)
  got = _same_structure(S.spgemm(S.from_scipy(a, DEV), S.from_scipy(b, DEV)), ref)
  # products of a cell are added in k-ascending order, the order of scipy's csr_matmat
  np.testing.assert_allclose(got.data, ref.data, rtol=4 * np.finfo(dtype).eps * 16, atol=1e-6)
  eye = sps.identity(500, dtype=dtype, format='csr')
  got = _same_structure(S.spgemm(S.from_scipy(a, DEV), S.from_scipy(eye, DEV)), a)
  np.testing.assert_array_equal(got.data, a.data)


def test_pagerank_sized_tile_properties():
  """configs-style full size: 900 000 pages x 10 out-links (tests/benchmark_pagerank.py:124-127).  Size-independent
  checks: W x ones == in-degree histogram (integer-valued: exact), transpose twice == identity, structure sorted."""
  torch = pytest.importorskip('torch')             # an independent calculator on the device (tests/dev.py)
  from tests.dev import T
  n, deg = 900000, 10
  rng = np.random.RandomState(1)
  rows_h = rng.randint(0, n, size=n * deg).astype(np.int32)
  rows = dev(rows_h)
  cols = dev(np.repeat(np.arange(n, dtype=np.int32), deg))
  vals = D.full((n * deg,), 1, np.float32)
  W = S.from_coo((n, n), np.float32, rows, cols, vals)
  assert int(W.indptr[-1]) == W.nnz and W.nnz <= n * deg
  assert float(W.data.sum()) == n * deg                       # duplicates were added, nothing lost
  indeg = torch.bincount(T(rows).long(), minlength=n).float()
  ones = D.full((n, 1), 1, np.float32)
  assert torch.equal(T(S.spmm(W, ones)).reshape(-1), indeg)
  assert torch.equal(T(S.row_sums(W)), indeg)
  Wt = S.transpose(W)
  assert torch.equal(T(S.spmm(Wt, ones)).reshape(-1), torch.full((n,), float(deg), device='cuda'))
  Wtt = S.transpose(Wt)
  assert all(torch.equal(T(p), T(q)) for p, q in ((Wtt.indptr, W.indptr), (Wtt.indices, W.indices), (Wtt.data, W.data)))
  # column indices ascend strictly inside every row
  ind, ptr = T(W.indices), T(W.indptr)
  d = ind[1:].long() - ind[:-1].long()
  row_start = torch.zeros(W.nnz, dtype=torch.bool, device='cuda')
  row_start[ptr[1:-1][ptr[1:-1] < W.nnz]] = True
  assert bool(torch.all((d > 0) | row_start[1:]))


def test_sparse_fuzz_against_scipy():
  """Random shapes (incl. empty rows / columns, single rows, no entries) through every structural and numeric entry
  point; integer values, so scipy's results must be reproduced exactly."""
  import os
  rng = np.random.RandomState(2024)
  n_cases = int(os.environ.get('SPARTAN_FUZZ_N', '60'))
  for case in range(n_cases):
    dtype = [np.float32, np.float64][rng.randint(2)]
    m, k, n = [int(rng.choice([1, 2, 7, 33, 200, 1025, 3000])) for _ in range(3)]
    dens = float(rng.choice([0.0, 0.002, 0.05, 0.4]))
    a = _canon(_rand_coo(rng, m, k, int(dens * m * k), dtype, dup=bool(rng.randint(2)), integer=True))
    b = _canon(_rand_coo(rng, k, n, int(float(rng.choice([0.0, 0.01, 0.2])) * k * n), dtype, integer=True))
    A, B = S.from_scipy(a, DEV), S.from_scipy(b, DEV)
    tag = 'case %d: %s [%d x %d] nnz %d, [%d x %d] nnz %d' % (case, np.dtype(dtype).name, m, k, a.nnz, k, n, b.nnz)
    got = _same_structure(S.transpose(A), _canon(a.T))
    np.testing.assert_array_equal(got.data, _canon(a.T).data, err_msg=tag)
    r0, r1 = sorted(rng.randint(0, m + 1, size=2))
    c0, c1 = sorted(rng.randint(0, k + 1, size=2))
    ref = _canon(a[r0:r1, c0:c1])
    got = _same_structure(S.slice_box(A, int(r0), int(r1), int(c0), int(c1)), ref)
    np.testing.assert_array_equal(got.data, ref.data, err_msg=tag)
    np.testing.assert_array_equal(S.to_scipy(S.add(A, S.transpose(S.transpose(A)), -1)).toarray(), np.zeros((m, k), dtype), err_msg=tag)
    x = rng.randint(-2, 3, size=(k, 1)).astype(dtype)
    np.testing.assert_array_equal(S.spmm(A, dev(x)).numpy(), a @ x, err_msg=tag)
    xm = rng.randint(-2, 3, size=(k, int(rng.choice([2, 5, 70])))).astype(dtype)
    np.testing.assert_array_equal(S.spmm(A, dev(xm)).numpy(), a @ xm, err_msg=tag)
    np.testing.assert_array_equal(S.row_sums(A).numpy(), np.asarray(a.sum(axis=1)).ravel(), err_msg=tag)
    np.testing.assert_array_equal(S.to_dense(A).numpy(), a.toarray(), err_msg=tag)
    prod = S.to_scipy(S.spgemm(A, B))
    np.testing.assert_array_equal(prod.toarray(), (a @ b).toarray(), err_msg=tag)
    assert prod.has_canonical_format


def _site_matrix(rng, n, deg, sites, local_frac, dtype=np.float32, integer=False):
  """benchmark_pagerank.py-style links: `deg` out-links per page, `local_frac` of them inside the page's site."""
  cols = np.repeat(np.arange(n, dtype=np.int64), deg)
  per = n // sites
  local = np.minimum((cols // per) * per + rng.randint(0, per, size=n * deg), n - 1)
  far = rng.randint(0, n, size=n * deg)
  rows = np.where(rng.rand(n * deg) <= local_frac, local, far)
  vals = rng.randint(-3, 4, size=n * deg).astype(dtype) if integer else rng.standard_normal(n * deg).astype(dtype)
  return _canon(sps.coo_matrix((vals, (rows, cols)), shape=(n, n)))


@pytest.mark.parametrize('n,deg,sites,local_frac', [(120000, 10, 4, 0.9),     # staged site slices + far links both sides
                                                    (70000, 6, 1, 1.0),       # every slice staged or none
                                                    (50000, 12, 16, 0.5),     # small sites: mostly direct segments
                                                    (8000, 40, 2, 0.8)])      # few rows, longer ones
def test_spmv_column_blocked_plan_is_bit_identical(monkeypatch, n, deg, sites, local_frac):
  """The column-blocked product (csrc/spmv_blocked.hip) against scipy and the planned stream kernel: same bits for
  random fp32 values -- a row's products are added in storage order by all three -- with and without `accumulate`.
  (Without any plan the stream kernel adds the rows that span two of its chunks as partial sums + carries.)"""
  rng = np.random.RandomState(n + deg)
  a = _site_matrix(rng, n, deg, sites, local_frac)
  A = S.from_scipy(a, DEV)
  assert S.spmv_block_plan(A) is not False
  x = rng.standard_normal((n, 1)).astype(np.float32)
  got = S.spmm(A, dev(x)).numpy()
  np.testing.assert_array_equal(got, a @ x)
  y = D.from_numpy(np.arange(n, dtype=np.float32).reshape(n, 1))
  S.spmm(A, dev(x), out=y, accumulate=True)
  np.testing.assert_array_equal(y.numpy(), np.arange(n, dtype=np.float32).reshape(n, 1) + (a @ x))
  v = dev(x.reshape(-1))
  np.testing.assert_array_equal(S.spmm(A, v).numpy(), (a @ x).reshape(-1))
  # the knob that keeps the stream kernel
  monkeypatch.setenv('SP_SPMV_BLOCKED', '0')
  B = S.from_scipy(a, DEV)
  assert S.spmv_block_plan(B) is False
  if int(np.diff(a.indptr).max()) <= 65:
    # (rows of more than 65 entries: the stream kernel adds per-chunk partial sums, not storage order)
    np.testing.assert_array_equal(S.spmm(B, dev(x)).numpy(), got)
  else:
    np.testing.assert_allclose(S.spmm(B, dev(x)).numpy(), got, rtol=1e-5, atol=1e-5)


def test_spmv_column_blocked_plan_refuses_unsorted_rows_and_small_tiles():
  rng = np.random.RandomState(5)
  a = _site_matrix(rng, 60000, 8, 2, 0.9, integer=True)
  A = S.from_scipy(a, DEV)
  # swap two columns inside one row of a copy of the tile: no longer canonical, the builder must say so
  ind = A.indices.numpy().copy()
  ptr = A.indptr.numpy()
  r = int(np.argmax(np.diff(ptr) >= 2))
  val = A.data.numpy().copy()
  ind[ptr[r]], ind[ptr[r] + 1] = ind[ptr[r] + 1], ind[ptr[r]]
  val[ptr[r]], val[ptr[r] + 1] = val[ptr[r] + 1], val[ptr[r]]
  B = S.CsrTile(A.shape, A.dtype, A.indptr, dev(ind), dev(val))
  assert S.spmv_block_plan(B) is False
  x = rng.randint(-2, 3, size=(60000, 1)).astype(np.float32)
  np.testing.assert_array_equal(S.spmm(B, dev(x)).numpy(), a @ x)       # integer values: order-free
  small = S.from_scipy(_site_matrix(rng, 3000, 5, 1, 1.0), DEV)
  assert S.spmv_block_plan(small) is False
  a64 = _site_matrix(rng, 60000, 8, 2, 0.9, dtype=np.float64)
  assert S.spmv_block_plan(S.from_scipy(a64, DEV)) is False


def test_spmv_column_blocked_plan_ragged_shapes():
  """Row and column counts that are multiples of nothing (a partly filled last row block, a last slice of x narrower
  than the others and staged), a band of empty rows, a rectangular tile, and a vector that is not 16-byte aligned
  (packed into an aligned buffer first): all bit-identical to scipy."""
  rng = np.random.RandomState(77)
  m, k, deg = 100003, 90001, 9
  cols = np.repeat(np.arange(k, dtype=np.int64), deg)
  rows = rng.randint(0, m, size=k * deg)
  rows[(rows >= 40000) & (rows < 43000)] = 39999               # rows 40000..42999 stay empty; one row gets very long?
  near = rng.rand(k * deg) < 0.85                              # most links of the last columns go to the last rows
  rows = np.where(near, np.minimum(m - 1, (cols * m) // k + rng.randint(-3000, 3000, size=k * deg)).clip(0), rows)
  rows[(rows >= 40000) & (rows < 43000)] = 43000
  a = _canon(sps.coo_matrix((rng.standard_normal(k * deg).astype(np.float32), (rows, cols)), shape=(m, k)))
  assert int(np.diff(a.indptr)[40000:43000].max()) == 0
  A = S.from_scipy(a, DEV)
  assert S.spmv_block_plan(A) is not False
  x = rng.standard_normal((k, 1)).astype(np.float32)
  np.testing.assert_array_equal(S.spmm(A, dev(x)).numpy(), a @ x)
  pad = dev(np.concatenate([np.zeros((1, 1), np.float32), x]))
  xv = pad[1:]                                                 # same values, 4 bytes off a 16-byte boundary
  assert xv.data_ptr() % 16 != 0
  want = a @ x
  got = S.spmm(A, xv).numpy()
  if int(np.diff(a.indptr).max()) <= 65:
    np.testing.assert_array_equal(got, want)
  else:
    # (row 43000 holds thousands of entries: the stream kernel adds it as per-chunk partial sums)
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-4)


def test_spmv_column_blocked_plan_strided_operands():
  """x as a column VIEW of a wider matrix (X[:, 1:2]: strides (4, 1) -- what Tile.get hands out), x as a vector with
  a step, and a caller's `out` that is a column of a wider matrix: the blocked kernel wants packed vectors
  (include/spartan_hip.h), so the host packs / pastes; values must be those of the packed operands."""
  rng = np.random.RandomState(91)
  n = 60000
  a = _site_matrix(rng, n, 8, 2, 0.9)
  A = S.from_scipy(a, DEV)
  assert S.spmv_block_plan(A) is not False
  wide = rng.standard_normal((n, 4)).astype(np.float32)
  W = dev(wide)
  col = W[:, 1:2]
  assert not col.is_contiguous()
  np.testing.assert_array_equal(S.spmm(A, col).numpy(), a @ wide[:, 1:2])
  stepped = dev(wide.reshape(-1))[::4]                          # column 0 as a 1-D view with a step
  np.testing.assert_array_equal(S.spmm(A, stepped).numpy(), a @ wide[:, 0])
  np.testing.assert_array_equal(S.spmm(A, stepped.reshape(-1, 1)).numpy(), a @ wide[:, 0:1])
  out_wide = dev(np.full((n, 3), 7, np.float32))
  target = out_wide[:, 2:3]
  got = S.spmm(A, col, out=target)
  assert got.data_ptr() == target.data_ptr()
  want = np.full((n, 3), 7, np.float32)
  want[:, 2:3] = a @ wide[:, 1:2]
  np.testing.assert_array_equal(out_wide.numpy(), want)
  S.spmm(A, col, out=target, accumulate=True)
  want[:, 2:3] = want[:, 2:3] + (a @ wide[:, 1:2])
  np.testing.assert_array_equal(out_wide.numpy(), want)
