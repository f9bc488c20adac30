"""The error window of the k-means split tier (spartan_amd/csrc/kmeans_split.hpp), checked on the host: the kernel's
arithmetic -- operands shifted by mu and rounded to fp32, cut into two bf16 numbers, three exact products per feature
accumulated in fp32 -- is restated in NumPy (bf16 by bit manipulation, one rounding to nearest per addend: the model
the bound grants the MFMA) and its scores are compared with exact ones.  What must hold for the labels to be exact:
|score_kernel - score_true| <= E / 2 for every (point, centre), E = u (F |x~| |c~|max + 2 |c~|max^2), F = 6.1 D + 1550,
scores halved as the kernel keeps them.  No GPU: this pins the derivation, tools/fuzz_kmeans.py the kernel."""
import zlib

import numpy as np
import pytest

U = 2.0 ** -24


def bf16(v):
  """Round-to-nearest-even fp32 -> bf16 (returned as fp32)."""
  b = np.asarray(v, np.float32).view(np.uint32).astype(np.uint64)
  r = ((b + 0x7fff + ((b >> 16) & 1)) >> 16) << 16
  return r.astype(np.uint32).view(np.float32)


def split(v):
  hi = bf16(v)
  mid = bf16(v.astype(np.float32) - hi)
  return hi, mid


def kernel_scores(x, c, mu, order):
  """Halved scores |c~|^2/2 - x~.c~ as the split tier computes them; `order` permutes the 3 D addends."""
  xs = (x.astype(np.float64) - mu).astype(np.float32)                 # fl32(x - mu)
  cs = (c.astype(np.float64) - mu).astype(np.float32)
  xh, xm = split(xs)
  ch, cm = split(cs)
  n, k, d = x.shape[0], c.shape[0], x.shape[1]
  out = np.empty((n, k), np.float32)
  chalf = (0.5 * (cs.astype(np.float64) ** 2).sum(axis=1)).astype(np.float32)
  for i in range(n):
    terms = np.concatenate([xm[i] * ch, xh[i] * cm, xh[i] * ch], axis=1)       # (k, 3 d): exact products in fp32
    assert np.array_equal(terms.astype(np.float64), np.concatenate([xm[i].astype(np.float64) * ch, xh[i].astype(np.float64) * cm,
                                                                    xh[i].astype(np.float64) * ch], axis=1))
    acc = np.zeros(k, np.float32)
    for j in order:
      acc = acc + terms[:, j]                                                     # one rounding to nearest per addend
    out[i] = chalf - acc
  return out, xs, cs


def true_scores(x, c, mu):
  xs = x.astype(np.float64) - mu
  cs = c.astype(np.float64) - mu
  return 0.5 * (cs ** 2).sum(axis=1)[None, :] - xs @ cs.T


@pytest.mark.parametrize('case', ['uniform', 'wide', 'signed', 'clustered', 'one_feature'])
@pytest.mark.parametrize('order', ['k', 'reversed', 'shuffled'])
def test_split_scores_stay_inside_the_window(case, order):
  rng = np.random.RandomState(zlib.crc32(('%s/%s' % (case, order)).encode()))
  n, k, d = 24, 40, (1 if case == 'one_feature' else 96)
  x = rng.rand(n, d)
  c = rng.rand(k, d)
  if case == 'wide':
    x *= 10.0 ** rng.randint(-5, 5, size=(n, d))
    c *= 10.0 ** rng.randint(-5, 5, size=(k, d))
  elif case == 'signed':
    x, c = x - 0.5, c - 0.5
  elif case == 'clustered':
    c = 0.5 + 0.01 * rng.randn(k, d)
    x = c[rng.randint(0, k, size=n)] + 1e-3 * rng.randn(n, d)
  x, c = x.astype(np.float32), c.astype(np.float32)
  mu = x.mean(axis=0).astype(np.float32).astype(np.float64)                   # any vector is a valid shift
  idx = np.arange(3 * d)
  if order == 'reversed':
    idx = idx[::-1]
  elif order == 'shuffled':
    rng.shuffle(idx)
  got, xs, cs = kernel_scores(x, c, mu, idx)
  want = true_scores(x, c, mu)
  F = 6.1 * d + 1550.0
  xn = np.sqrt((xs.astype(np.float64) ** 2).sum(axis=1))
  cmax2 = (cs.astype(np.float64) ** 2).sum(axis=1).max()
  E = U * (F * xn * np.sqrt(cmax2) + 2.0 * cmax2)                             # per point
  err = np.abs(got.astype(np.float64) - want)
  assert np.all(err <= 0.5 * E[:, None]), (err / E[:, None]).max()
  # and the pieces of the derivation: the terms left out, and the accumulation
  S = np.abs(xs.astype(np.float64))[:, None, :] * np.abs(cs.astype(np.float64))[None, :, :]
  xh, xm = split(xs)
  ch, cm = split(cs)
  kept = (xh.astype(np.float64) @ ch.astype(np.float64).T + xh.astype(np.float64) @ cm.astype(np.float64).T +
          xm.astype(np.float64) @ ch.astype(np.float64).T)
  left_out = np.abs(xs.astype(np.float64) @ cs.astype(np.float64).T - kept)
  assert np.all(left_out <= 3.0001 * 2.0 ** -16 * S.sum(axis=2) + 1e-300)
  assert np.all(np.abs(xs - (xh + xm)) <= 2.0 ** -16 * np.abs(xs) + 1e-45)


def test_the_exact_argmin_is_always_inside_the_candidate_window():
  """The decision rule: a point is settled by the filter only if second - best > 2 E (halved scores), and the re-check
  marks every centre whose score is within E of the best one.  With |error| <= E / 2 per score, the true nearest
  centre of every point is then either the filter's verdict or among its candidates."""
  rng = np.random.RandomState(7)
  n, k, d = 60, 50, 64
  c = (0.5 + 0.02 * rng.randn(k, d)).astype(np.float32)
  x = (c[rng.randint(0, k, size=n)] + 2e-3 * rng.randn(n, d)).astype(np.float32)
  mu = x.mean(axis=0).astype(np.float32).astype(np.float64)
  got, xs, cs = kernel_scores(x, c, mu, np.arange(3 * d))
  F = 6.1 * d + 1550.0
  xn = np.sqrt((xs.astype(np.float64) ** 2).sum(axis=1))
  cmax2 = (cs.astype(np.float64) ** 2).sum(axis=1).max()
  E = U * (F * xn * np.sqrt(cmax2) + 2.0 * cmax2)
  exact = np.argmin(((x.astype(np.float64)[:, None, :] - c.astype(np.float64)[None, :, :]) ** 2).sum(axis=2), axis=1)
  order = np.sort(got, axis=1)
  best, second = order[:, 0], order[:, 1]
  for i in range(n):
    if 2.0 * (second[i] - best[i]) > 4.0 * E[i]:
      assert np.argmin(got[i]) == exact[i]
    else:
      assert got[i, exact[i]] <= best[i] + E[i]
