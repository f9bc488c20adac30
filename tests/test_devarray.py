"""View arithmetic of spartan_amd.devarray.DevArray against NumPy's own (no GPU: HostStorage keeps the bytes in
host memory; only shapes / strides / offsets are computed by the code under test)."""
import numpy as np
import pytest

from spartan_amd import devarray as D


def host(a):
  return D.from_numpy(a, storage_cls=D.HostStorage)


def _random_index(rng, shape):
  idx = []
  for n in shape:
    kind = rng.randint(0, 5)
    if kind == 0:
      idx.append(slice(None))
    elif kind == 1 and n > 0:
      idx.append(int(rng.randint(-n, n)))
    elif kind == 2:
      idx.append(None)
      idx.append(slice(None))
    else:
      a, b = sorted(rng.randint(-n - 1, n + 2, size=2))
      step = int(rng.choice([1, 1, 2, 3, -1, -2]))
      idx.append(slice(int(a), int(b), step) if step > 0 else slice(int(b), int(a), step))
  if rng.randint(0, 4) == 0 and len(idx) > 1:
    cut = rng.randint(0, len(idx))
    idx = idx[:cut] + [Ellipsis]
  return tuple(idx)


def test_views_match_numpy():
  rng = np.random.RandomState(7)
  for case in range(400):
    nd = rng.randint(1, 5)
    shape = tuple(int(s) for s in rng.randint(1, 7, size=nd))
    a = np.arange(int(np.prod(shape)), dtype=[np.float32, np.int64, np.uint8, np.float64][case % 4]).reshape(shape)
    d = host(a)
    for step in range(4):
      op = rng.randint(0, 4)
      if op == 0:
        idx = _random_index(rng, a.shape)
        a, d = a[idx], d[idx]
      elif op == 1 and a.ndim >= 2:
        perm = tuple(int(p) for p in rng.permutation(a.ndim))
        a, d = a.transpose(perm), d.permute(*perm)
      elif op == 2 and a.ndim >= 2:
        s, t = int(rng.randint(0, a.ndim)), int(rng.randint(0, a.ndim))
        a, d = np.moveaxis(a, s, t), d.movedim(s, t)
      elif op == 3 and a.size:
        # a reshape NumPy can do without a copy must be a view here too
        n = a.size
        divs = [k for k in range(1, n + 1) if n % k == 0]
        k = int(rng.choice(divs))
        new = (k, n // k) if rng.randint(0, 2) else (n // k, 1, k)
        try:
          v = a.view()
          v.shape = new                       # raises if NumPy needs a copy
        except AttributeError:
          continue
        a, d = v, d.reshape(new)
      assert d.shape == a.shape, (case, step)
      assert d.is_contiguous() == (a.flags['C_CONTIGUOUS'] or a.size == 0), (case, step, a.shape, a.strides, d.strides)
      np.testing.assert_array_equal(d.numpy(), a, err_msg=str((case, step)))
      if a.size > 1:
        assert [s * a.itemsize for n, s in zip(d.shape, d.strides) if n > 1] == \
            [s for n, s in zip(a.shape, a.strides) if n > 1], (case, step)


def test_reshape_of_a_strided_view_that_needs_a_copy_is_detected():
  a = np.arange(24, dtype=np.float32).reshape(4, 6)
  d = host(a)
  assert D._view_reshape(d.t().shape, d.t().strides, (24,)) is None
  assert D._view_reshape(d[:, :3].shape, d[:, :3].strides, (12,)) is None
  assert D._view_reshape(d[:, :3].shape, d[:, :3].strides, (2, 2, 3)) == (12, 6, 1)
  assert D._view_reshape((4, 6), (6, 1), (2, 2, 3, 2)) == (12, 6, 2, 1)
  assert D._view_reshape((4, 1, 6), (6, 99, 1), (24,)) == (1,)
  np.testing.assert_array_equal(d[1:3].reshape(-1).numpy(), a[1:3].reshape(-1))
  np.testing.assert_array_equal(d.reshape(2, -1).numpy(), a.reshape(2, -1))
  with pytest.raises(ValueError):
    d.reshape(5, 5)


def test_scalars_and_edge_cases():
  a = np.arange(12, dtype=np.int64).reshape(3, 4)
  d = host(a)
  assert d[1, 2].shape == () and d[1, 2].item() == 6
  assert d[-1][-1].item() == 11
  assert len(d) == 3 and d.ndim == 2 and d.size == 12 and d.nbytes == 96 and d.itemsize == 8
  assert d.T.shape == (4, 3) and d.t().stride() == (1, 4) and d.stride(0) == 4 and d.dim() == 2 and d.numel() == 12
  assert d[0:0].shape == (0, 4) and d[0:0].is_contiguous()
  assert d.data_ptr() + 8 * 5 == d[1, 1].data_ptr()
  z = host(np.float32(3.5))
  assert z.shape == () and z.item() == 3.5 and z.reshape(1).shape == (1,)
  with pytest.raises(IndexError):
    d[3]
  with pytest.raises(IndexError):
    d[0, 0, 0]
  with pytest.raises(TypeError):
    d[np.array([0, 1])]
  np.testing.assert_array_equal(np.asarray(d), a)
  np.testing.assert_array_equal(d.cpu().numpy(), a)
  np.testing.assert_array_equal(d.squeeze().numpy(), a)
  np.testing.assert_array_equal(d[None, :, None].squeeze().numpy(), a)
  np.testing.assert_array_equal(d.swapaxes(0, 1).numpy(), a.swapaxes(0, 1))


def test_device_tile_cannot_is_the_only_fallback_signal():
  """What a device tile cannot do is said with ONE exception type, raised by DevArray alone: that (and nothing a user
  function raises itself) is what sends a local function to host copies (backend_hip.call_local_fn)."""
  d = host(np.arange(12, dtype=np.float32).reshape(3, 4))
  with pytest.raises(D.DeviceTileCannot):
    np.cumsum(d, axis=1)                    # a NumPy function without a kernel behind it
  with pytest.raises(D.DeviceTileCannot):
    d[np.array([0, 2])]                     # index arrays
  with pytest.raises(D.DeviceTileCannot):
    d.cumsum                                # an ndarray attribute without a device form
  with pytest.raises(AttributeError) as info:
    d.no_such_method                        # a typo is the caller's error, not a reason to fall back to the host
  assert not isinstance(info.value, D.DeviceTileCannot)
  with pytest.raises(D.DeviceTileCannot):
    np.add.accumulate(d)                    # a ufunc method without a kernel
  assert not hasattr(d, 'no_such_method') and hasattr(d, 'reshape')
  assert issubclass(D.DeviceTileCannot, TypeError) and issubclass(D.DeviceTileCannot, AttributeError)


def test_call_local_fn_runs_a_failing_user_function_once():
  """A bug in the user's function is the user's error: it propagates from the first run, the function is not run
  again on host copies (a function with side effects would not be idempotent)."""
  from spartan_amd import backend_hip
  be = backend_hip.HipBackend.__new__(backend_hip.HipBackend)      # (no device: only the dispatch is under test)
  be.launches = be.host_round_trips = 0
  be._warned_host = set()
  d = host(np.ones((2, 2), np.float32))
  calls = []

  def buggy(x):
    calls.append(1)
    return x.shape[5]                        # IndexError of the function's own

  with pytest.raises(IndexError):
    be.call_local_fn(buggy, [d], {})
  assert len(calls) == 1 and be.host_round_trips == 0

  def needs_host(x):
    calls.append(2)
    return np.cumsum(x, axis=0) if isinstance(x, np.ndarray) else x.cumsum(0)

  with pytest.warns(RuntimeWarning, match='host copies'):
    try:
      be.call_local_fn(needs_host, [d], {})
    except Exception:                        # (uploading the result needs the device library: not under test here)
      pass
  assert calls.count(2) == 2 and be.host_round_trips == 1


def test_sum_prod_accumulator_dtypes_follow_numpy():
  """The dtype table of ndarray.sum / prod (bool, int32 -> int64; floats and int64 unchanged); an unsigned tile
  would accumulate in uint64, which no device tile can hold: the sentinel, not a silently different dtype."""
  for dt in (np.bool_, np.int32, np.int64, np.float32, np.float64):
    d = host(np.ones((2, 3), dt))
    for what in ('sum', 'prod'):
      want = getattr(np.ones((2, 3), dt), what)().dtype
      acc = d._accumulator(None, what)
      assert np.dtype(acc if acc is not None else dt) == want, (dt, what)
  with pytest.raises(D.DeviceTileCannot):
    host(np.ones((2, 3), np.uint8))._accumulator(None, 'sum')
  assert host(np.ones((2, 3), np.uint8))._accumulator(np.int64, 'sum') == np.int64
