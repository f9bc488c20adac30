"""Lowering of fused LocalExpr trees to sp_program (the HIP kernel bytecode).

The reference's precedent for a compiled local-op backend is ParakeetExpr /
the ParakeetGeneration pass (spartan/expr/operator/local.py:187-209,
optimize.py:321-370): a MapExpr's `op` tree is turned into one compiled
function when every node is understood, and code generation failure is a soft
error there.  Here the set of understood callables is a registry
(`register_map_rule` / `register_reduce_rule`); any OTHER Python callable handed to
`map` is run once on `Traced` stand-ins for its tiles, so a user function made of
operators and NumPy ufuncs (`lambda t: np.sqrt(t * 2 + 1)`) becomes part of the fused
kernel as well; what cannot be traced (indexing, reductions, data-dependent `if`)
raises `NotLowerable` -- loudly, there is no CPU fallback.

NumPy typing is reproduced exactly: every node's dtype is obtained by applying
the very same ufunc to zero-dimensional dummies of the operand dtypes (so
promotion, comparison->bool, int/int->float64 ... are NumPy's own rules), and
Python scalars stay weak (NEP 50).
"""
import numpy as np

from . import _hip
from .array import distarray, tile
from .expr import builtins as B
from .expr.local import FnCallExpr, LocalInput, LocalMapLocationExpr
from .program import (Program, ProgramTooLarge, broadcast_strides, class_of, collapse, dense_strides,
                      join_class)


class NotLowerable(TypeError):
  """The local function has no registered GPU lowering."""



# --------------------------------------------------------------------- values
class V(object):
  """A typed value in the tree being lowered."""
  __slots__ = ('kind', 'dtype', 'shape', 'tensor', 'value', 'op', 'args', 'weak')

  def __init__(self, kind, dtype=None, shape=(), tensor=None, value=None, op=None, args=(), weak=False):
    self.kind = kind      # 'tensor' | 'const' | 'shape' | 'op' | 'iota' | 'extent' | 'axis'
    self.dtype = None if dtype is None else np.dtype(dtype)
    self.shape = tuple(int(s) for s in shape)
    self.tensor = tensor
    self.value = value
    self.op = op
    self.args = list(args)
    self.weak = weak


def const(value, dtype=None):
  """A scalar constant; Python scalars are weak (dtype decided by the other operand)."""
  if dtype is not None:
    return V('const', dtype=dtype, value=np.dtype(dtype).type(value).item(), weak=False)
  if isinstance(value, np.generic):
    return V('const', dtype=value.dtype, value=value.item(), weak=False)
  if isinstance(value, bool):
    return V('const', dtype=np.bool_, value=value, weak=True)
  if isinstance(value, int):
    return V('const', dtype=np.int64, value=value, weak=True)
  if isinstance(value, float):
    return V('const', dtype=np.float64, value=value, weak=True)
  raise NotLowerable('cannot use %r as a kernel constant' % (value,))


def _dummy(v):
  if v.kind == 'const' and v.weak:
    return v.value
  return np.zeros((), dtype=v.dtype)


_result_dtypes = {}


def _weak_key(value, by_value=False):
  """What a weak Python scalar contributes to a result dtype (NEP 50): its kind -- its value only where it could
  overflow the other operand's integer type (by_value: an operand narrower than int32, unsigned or bool is in the
  operation: uint8 + 300 and uint8 + (-1) are refused by NumPy, uint8 + 1 is uint8)."""
  t = type(value)
  if t is float or t is bool or (t is int and not by_value and -2147483648 <= value < 2147483648):
    return ('weak', t)
  return (t, value)


def _narrow_int(dtype):
  dt = np.dtype(dtype)
  return dt.kind in 'ub' or (dt.kind == 'i' and dt.itemsize < 4)


def warm_result_dtypes():
  """Ask NumPy once, when the backend comes up, for the result dtype of every registered ufunc over the operand
  types programs are made of (fp32 / fp64 / int64 / int32 / bool tiles, weak Python floats and ints): the first
  lowering of a program then meets a filled table instead of paying ~30 us per operator it has not seen."""
  tiles = [np.dtype(t) for t in (np.float32, np.float64, np.int64, np.int32, np.bool_)]
  weak = [1.5, 1]
  with np.errstate(all='ignore'):
    for fn in list(MAP_RULES):
      if not isinstance(fn, np.ufunc):
        continue
      combos = []
      if fn.nin == 1:
        combos = [(t,) for t in tiles]
      elif fn.nin == 2:
        combos = [(a, b) for a in tiles for b in tiles if a == b or a.kind != b.kind]
        combos += [(a, w) for a in tiles for w in weak] + [(w, a) for a in tiles for w in weak]
      for combo in combos:
        key = (fn, tuple([_weak_key(c) if not isinstance(c, np.dtype) else c for c in combo]))
        if key in _result_dtypes:
          continue
        try:
          res = fn(*[np.zeros((), c) if isinstance(c, np.dtype) else c for c in combo])
          _result_dtypes[key] = np.asarray(res).dtype
        except Exception:   # noqa: BLE001  (a combination NumPy refuses is refused again when a program asks)
          pass


def _bshape(*shapes):
  return tuple(np.broadcast_shapes(*shapes))


def apply(opname, np_fn, args):
  """Typed application of an elementwise op; result dtype = NumPy's for np_fn."""
  if all(a.kind == 'const' for a in args):
    with np.errstate(all='ignore'):
      r = np_fn(*[a.value if a.weak else np.dtype(a.dtype).type(a.value) for a in args])
    # a ufunc applied to Python scalars returns a NumPy scalar, which is STRONGLY typed from then on
    # (np.abs(2) + 1 is np.int64(3); fp32 % np.int64(3) is float64) -- the same value the unfused evaluation of
    # this sub-tree produces as a 0-d tile
    return const(np.asarray(r)[()])
  # NumPy's own answer, asked once per (function, operand types): the dummy call costs tens of microseconds
  # (np.errstate alone ~10), a fused tree has one per operator, and a driver loop lowers the same trees for ever
  try:
    by_value = any(_narrow_int(a.dtype) for a in args if not (a.kind == 'const' and a.weak))
    key = (np_fn, tuple([_weak_key(a.value, by_value) if (a.kind == 'const' and a.weak) else a.dtype for a in args]))
    dt = _result_dtypes.get(key)
  except TypeError:
    key = dt = None
  if dt is None:
    with np.errstate(all='ignore'):
      res = np_fn(*[_dummy(a) for a in args])
    dt = np.asarray(res).dtype
    if key is not None:
      if len(_result_dtypes) > 4096:
        _result_dtypes.clear()
      _result_dtypes[key] = dt
  shapes = [a.shape for a in args]
  shape = shapes[0] if all(s == shapes[0] or s == () for s in shapes[1:]) and shapes[0] != () else _bshape(*shapes)
  return V('op', dtype=dt, shape=shape, op=opname, args=args)


def cast(v, dtype):
  """ndarray.astype(dtype)."""
  dtype = np.dtype(dtype)
  if v.kind == 'const':
    return const(np.asarray(v.value).astype(dtype)[()])
  if v.dtype == dtype:
    return v
  return V('op', dtype=dtype, shape=v.shape, op='CAST', args=[v])


# ---------------------------------------------------------------- rule tables
MAP_RULES = {}
REDUCE_RULES = {}


def register_map_rule(fn, rule):
  """rule(args: [V], kw: dict, ex) -> V"""
  MAP_RULES[fn] = rule


def register_reduce_rule(fn, rule):
  """rule(data: V, axis, ex) -> (red_op, V to reduce, natural result dtype)"""
  REDUCE_RULES[fn] = rule


def _ufunc(opname, fn):
  def rule(args, kw, ex):
    if kw:
      raise NotLowerable('keyword arguments %s of %s are not supported on the GPU' % (list(kw), fn))
    return apply(opname, fn, args)
  register_map_rule(fn, rule)


for _name, _fn in [
    ('ADD', np.add), ('SUB', np.subtract), ('MUL', np.multiply), ('DIV', np.divide),
    ('FLOORDIV', np.floor_divide), ('MOD', np.mod), ('FMOD', np.fmod), ('POW', np.power),
    ('MAX', np.maximum), ('MIN', np.minimum), ('EQ', np.equal), ('NE', np.not_equal), ('LT', np.less),
    ('LE', np.less_equal), ('GT', np.greater), ('GE', np.greater_equal), ('LAND', np.logical_and),
    ('LOR', np.logical_or), ('LXOR', np.logical_xor), ('LNOT', np.logical_not), ('NEG', np.negative),
    ('ABS', np.abs), ('SQRT', np.sqrt), ('SQUARE', np.square), ('EXP', np.exp), ('LOG', np.log),
    ('RECIP', np.reciprocal), ('SIGN', np.sign), ('FLOOR', np.floor), ('CEIL', np.ceil),
    ('TANH', np.tanh)]:
  _ufunc(_name, _fn)
# np.true_divide is np.divide, np.remainder is np.mod, np.absolute is np.abs in NumPy 2
try:   # statistics.py:224-225: norm_cdf maps scipy.stats.norm.cdf over the tiles
  import scipy.stats as _scipy_stats
  _ufunc('NORM_CDF', _scipy_stats.norm.cdf)
except ImportError:   # pragma: no cover
  _scipy_stats = None


def _where(args, kw, ex):
  c, a, b = args
  with np.errstate(all='ignore'):
    dt = np.asarray(np.where(True, _dummy(a), _dummy(b))).dtype
  return V('op', dtype=dt, shape=_bshape(c.shape, a.shape, b.shape), op='WHERE', args=[c, a, b])


register_map_rule(np.where, _where)


def _like_fill(value):
  def rule(args, kw, ex):
    src = args[0]
    return V('op', dtype=src.dtype, shape=src.shape, op='FILL', args=[const(value, src.dtype)])
  return rule


register_map_rule(B._make_zeros, _like_fill(0))
register_map_rule(B._make_ones, _like_fill(1))


def _full(args, kw, ex):
  src = args[0]
  dt = np.dtype(kw.get('dtype') or src.dtype)
  return V('op', dtype=dt, shape=src.shape, op='FILL', args=[const(kw['fill_value'], dt)])


register_map_rule(B._full_mapper, _full)


def _astype(args, kw, ex):
  return cast(args[0], kw['dtype'])


register_map_rule(B._astype_mapper, _astype)


def _arange(args, kw, ex):
  """creation.py:134-141: value = (ravelled_pos(ul, array_shape) + local index) * step + start."""
  from .array import extent as extent_mod
  src = args[0]
  dt = np.dtype(kw.get('dtype') or float)
  step, start = kw['step'], kw['start']
  iot = V('iota', dtype=np.int64, shape=src.shape)
  if B.ravel_contiguous(ex.ul, ex.lr, ex.array_shape):
    # np.arange(ex_start, ex_stop, step, dtype): element i = ex_start + i*step,
    # computed in the (float64 / int64) type of the Python arguments, then cast
    pos = extent_mod.ravelled_pos(ex.ul, ex.array_shape)
    v = apply('MUL', np.multiply, [iot, const(step)])
    v = apply('ADD', np.add, [v, const(pos * step + start)])
    return cast(v, dt)
  # a column / block tile (builtins._arange_mapper): position of every element in the whole array, from the
  # tile-local index digit by digit
  where, rest, gstride = None, iot, 1
  for axis in reversed(range(len(ex.ul))):
    n = ex.lr[axis] - ex.ul[axis]
    digit = apply('MOD', np.mod, [rest, const(n)]) if axis else rest
    rest = apply('FLOORDIV', np.floor_divide, [rest, const(n)]) if axis else rest
    term = apply('MUL', np.multiply, [apply('ADD', np.add, [digit, const(int(ex.ul[axis]))]), const(int(gstride))])
    where = term if where is None else apply('ADD', np.add, [where, term])
    gstride *= ex.array_shape[axis]
  v = apply('ADD', np.add, [apply('MUL', np.multiply, [where, const(step)]), const(start)])
  return cast(v, dt)


register_map_rule(B._arange_mapper, _arange)


def _eye(args, kw, ex):
  src = args[0]
  dt = np.dtype(kw.get('dtype') or float)
  ncols = src.shape[1]
  iot = V('iota', dtype=np.int64, shape=src.shape)
  row = apply('FLOORDIV', np.floor_divide, [iot, const(ncols)])
  col = apply('MOD', np.mod, [iot, const(ncols)])
  k = ex.ul[0] - ex.ul[1] + kw['k']
  hit = apply('EQ', np.equal, [col, apply('ADD', np.add, [row, const(k)])])
  return cast(hit, dt)


register_map_rule(B._eye_mapper, _eye)


def _arg_candidates(args, kw, ex):
  idx, val, best = args
  eq = apply('EQ', np.equal, [val, best])
  return V('op', dtype=np.int64, shape=_bshape(idx.shape, val.shape, best.shape), op='WHERE',
           args=[eq, idx, const(np.int64(kw['sentinel']))])


register_map_rule(B._arg_candidates, _arg_candidates)


def _sum_dtype(dt):
  dt = np.dtype(dt)
  if dt.kind in 'bi':
    return np.dtype(np.int64)
  if dt.kind == 'u':
    return np.dtype(np.uint64) if dt.itemsize == 8 else np.dtype(np.int64)
  return dt


register_reduce_rule(B._sum_local, lambda data, axis, ex: ('SUM', data, _sum_dtype(data.dtype)))
register_reduce_rule(B._prod_local, lambda data, axis, ex: ('PROD', data, _sum_dtype(data.dtype)))
register_reduce_rule(B._max_local, lambda data, axis, ex: ('MAX', data, data.dtype))
register_reduce_rule(B._min_local, lambda data, axis, ex: ('MIN', data, data.dtype))
register_reduce_rule(B._all_reducer, lambda data, axis, ex: ('AND', data, np.dtype(np.bool_)))
register_reduce_rule(B._any_reducer, lambda data, axis, ex: ('OR', data, np.dtype(np.bool_)))


def _count_nonzero(data, axis, ex):
  # sorting.py:126-133: count_nonzero for axis=None, (data > 0).sum(axis) otherwise
  if axis is None:
    return ('SUM', apply('NE', np.not_equal, [data, const(0)]), np.dtype(np.int64))
  return ('SUM', apply('GT', np.greater, [data, const(0)]), np.dtype(np.int64))


def _count_zero(data, axis, ex):
  return ('SUM', apply('EQ', np.equal, [data, const(0)]), np.dtype(np.int64))


register_reduce_rule(B._countnonzero_local, _count_nonzero)
register_reduce_rule(B._countzero_local, _count_zero)
# norm (statistics.py:203-215): np.abs(data).sum(axis) and np.square(data).sum(axis) as one fused map -> reduce
from .expr import manip as _M  # noqa: E402
register_reduce_rule(_M._abs_sum_local, lambda data, axis, ex: ('SUM', apply('ABS', np.abs, [data]), _sum_dtype(data.dtype)))
register_reduce_rule(_M._square_sum_local, lambda data, axis, ex: ('SUM', apply('SQUARE', np.square, [data]), _sum_dtype(data.dtype)))


# ------------------------------------------------------------------- inference
def value_of_input(x):
  """Wrap a fetched local value (backend tensor, EmptyBlob, NumPy data, scalar)."""
  if isinstance(x, tile.EmptyBlob):
    return V('shape', dtype=x.dtype, shape=x.shape)
  if isinstance(x, (bool, int, float)) and not isinstance(x, np.generic):
    return const(x)
  if isinstance(x, np.generic):
    return const(x)
  if isinstance(x, np.ndarray):
    if x.ndim == 0:
      return const(x[()])
    return V('tensor', dtype=x.dtype, shape=x.shape, tensor=x)  # uploaded by the backend
  if isinstance(x, distarray.Absent):
    raise AssertionError('Absent data reached a kernel on the executing rank')
  return V('tensor', dtype=None, shape=tuple(x.shape), tensor=x)  # dtype filled by the backend


def infer(op, inputs, ex, dtype_of):
  """LocalExpr tree -> V tree.  `inputs`: var name -> local value."""
  if isinstance(op, LocalInput):
    if op.idx == 'extent':
      return V('extent', value=ex)
    if op.idx == 'axis':
      return V('axis', value=inputs.get('axis'))
    x = inputs[op.idx]
    v = x if isinstance(x, V) else value_of_input(x)
    if v.kind == 'tensor' and v.dtype is None:
      v.dtype = np.dtype(dtype_of(v.tensor))
    return v
  if not isinstance(op, FnCallExpr):
    raise NotLowerable('cannot lower local expression %r' % (op,))
  rule = MAP_RULES.get(op.fn)
  args = [infer(d, inputs, ex, dtype_of) for d in op.deps]
  if rule is None:
    return trace(op, args, ex)
  args = [a for a in args if a.kind != 'extent']
  return rule(args, dict(op.kw), ex)


# ----------------------------------------------------------- tracing of user functions
class Traced(object):
  """Stand-in for a tile handed to a user's element-wise Python function (`map(x, lambda t: t * 2 + 1)`,
  tests/test_slice.py:23-24,41-56 of the reference).  Operators and NumPy ufuncs applied to it extend the
  tree being lowered, so the function becomes part of the fused HIP kernel; anything that is not
  element-wise (indexing, reductions, data-dependent `if`) is refused loudly."""
  __array_priority__ = 1000.0
  __slots__ = ('v',)

  def __init__(self, v):
    self.v = v

  shape = property(lambda self: self.v.shape)
  dtype = property(lambda self: self.v.dtype)
  ndim = property(lambda self: len(self.v.shape))
  size = property(lambda self: int(np.prod(self.v.shape, dtype=np.int64)))

  def astype(self, dtype):
    return Traced(cast(self.v, dtype))

  def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
    if method != '__call__' or kwargs.get('out') is not None:
      raise NotLowerable('only plain element-wise ufunc calls can be traced into a kernel (%s.%s)' % (ufunc.__name__, method))
    rule = MAP_RULES.get(ufunc)
    if rule is None:
      raise NotLowerable('no GPU lowering registered for %s' % ufunc.__name__)
    return Traced(rule([_as_value(i) for i in inputs], dict(kwargs), None))

  def __array_function__(self, func, types, args, kwargs):
    """Non-ufunc NumPy functions with a registered lowering (np.where ...)."""
    rule = MAP_RULES.get(func)
    if rule is None:
      raise NotLowerable('numpy.%s of a tile cannot be traced into an element-wise kernel' % getattr(func, '__name__', func))
    return Traced(rule([_as_value(i) for i in args], dict(kwargs), None))

  def _refuse(self, what):
    raise NotLowerable('%s inside a mapped function cannot be traced into an element-wise kernel' % what)

  def __getitem__(self, idx): self._refuse('indexing a tile')
  def __bool__(self): self._refuse('a data-dependent Python condition')
  def __iter__(self): self._refuse('iterating over a tile')
  def __len__(self): self._refuse('len() of a tile')
  def sum(self, *a, **k): self._refuse('a reduction (.sum)')
  def max(self, *a, **k): self._refuse('a reduction (.max)')
  def min(self, *a, **k): self._refuse('a reduction (.min)')
  def mean(self, *a, **k): self._refuse('a reduction (.mean)')
  def dot(self, *a, **k): self._refuse('a contraction (.dot)')
  def reshape(self, *a, **k): self._refuse('reshaping a tile')
  __hash__ = None


def _as_value(x):
  return x.v if isinstance(x, Traced) else value_of_input(x)


def _traced_op(ufunc, reflected=False):
  def method(self, other=None):
    ins = (self,) if other is None else ((other, self) if reflected else (self, other))
    return self.__array_ufunc__(ufunc, '__call__', *ins)
  return method


for _dunder, _uf in dict(add=np.add, sub=np.subtract, mul=np.multiply, truediv=np.divide, floordiv=np.floor_divide,
                         mod=np.mod, pow=np.power, lt=np.less, le=np.less_equal, gt=np.greater, ge=np.greater_equal,
                         eq=np.equal, ne=np.not_equal, and_=np.logical_and, or_=np.logical_or,
                         xor=np.logical_xor).items():
  _n = _dunder.rstrip('_')
  setattr(Traced, '__%s__' % _n, _traced_op(_uf))
  if _n in ('add', 'sub', 'mul', 'truediv', 'floordiv', 'mod', 'pow', 'and', 'or', 'xor'):
    setattr(Traced, '__r%s__' % _n, _traced_op(_uf, reflected=True))
Traced.__neg__ = _traced_op(np.negative)
Traced.__abs__ = _traced_op(np.abs)
Traced.__invert__ = _traced_op(np.logical_not)


def trace(op, args, ex):
  """Run the user's function once on Traced stand-ins to obtain its element-wise tree."""
  call = []
  for a in args:
    if a.kind == 'extent':
      call.append(ex.to_tuple())          # map_with_location hands the tile's position over (local.py:137-149)
    elif a.kind == 'const':
      call.append(a.value)
    else:
      call.append(Traced(a))
  try:
    res = op.fn(*call, **dict(op.kw or {}))
  except NotLowerable:
    raise
  except Exception as e:   # noqa: BLE001
    raise NotLowerable('local function %s could not be traced into an element-wise kernel (%s: %s); the HIP '
                       'backend has no CPU fallback -- use ufuncs / operators on the tile, or register a rule '
                       '(spartan_amd/lower.py: register_map_rule)' % (op.fn_name(), type(e).__name__, e))
  if isinstance(res, Traced):
    return res.v
  try:
    return value_of_input(res)
  except Exception:   # noqa: BLE001
    raise NotLowerable('local function %s returned %r, not an element-wise expression of its tiles'
                       % (op.fn_name(), type(res)))


# -------------------------------------------------------------------- emission
_NORMALISE = {np.dtype(np.float32): 'TO_F32', np.dtype(np.int32): 'TO_I32', np.dtype(np.uint8): 'TO_U8'}
# operator -> (name when the RIGHT operand is the constant, name when the LEFT one is): reg[b] (op) consts[a]
_WITH_CONST = {'ADD': ('ADDC', 'ADDC'), 'SUB': ('SUBC', 'RSUBC'), 'MUL': ('MULC', 'MULC'), 'DIV': ('DIVC', 'RDIVC'),
               'MAX': ('MAXC', 'MAXC'), 'MIN': ('MINC', 'MINC')}
_NO_NORMALISE_OPS = set(['EQ', 'NE', 'LT', 'LE', 'GT', 'GE', 'LAND', 'LOR', 'LXOR', 'LNOT', 'WHERE'])


def _classes(v, acc):
  if v.kind in ('tensor', 'op', 'iota') or (v.kind == 'const' and not v.weak):
    if v.dtype != np.bool_:       # 0/1 is exact in every class
      acc.append(class_of(v.dtype))
  elif v.kind == 'const' and v.weak and isinstance(v.value, float):
    acc.append(None)  # weak float: at least a float class, decided below
  for a in v.args:
    _classes(a, acc)


def choose_class(root, extra=()):
  acc = list(extra)
  _classes(root, acc)
  concrete = [c for c in acc if c is not None]
  cls = concrete[0] if concrete else _hip.SP_F64
  for c in concrete[1:]:
    cls = join_class(cls, c)
  if None in acc and cls == _hip.SP_I64:
    cls = _hip.SP_F64
  return cls


def _same_storage(a, b):
  """Two fetches of the same tile region are distinct view objects over the
  same HBM bytes: load them once (the reference loads each use separately)."""
  if a is b:
    return True
  try:
    return (a.data_ptr() == b.data_ptr() and a.dtype == b.dtype and tuple(a.shape) == tuple(b.shape) and
            tuple(a.stride()) == tuple(b.stride()))
  except AttributeError:
    return False


class Emitter(object):
  def __init__(self, cls, out_shape, contiguous=None):
    """contiguous: backend callable making a dense copy of a tensor the kernels cannot address in place."""
    self.cls = cls
    self.contiguous = contiguous
    self.out_shape = tuple(out_shape)
    self.prog = Program()
    self.tensors = []      # backend tensors, in input order
    self.in_vals = []      # (V, strides) per input
    self.free = []
    self.next_temp = None
    # common subexpressions (finish): structural key -> small integer, uses left per key, key <-> register holding it
    self.share = False
    self._key_of_node = {}
    self._key_ids = {}
    self._uses = {}
    self._reg_of_key = {}
    self._key_of_reg = {}
    self._same_as = {}

  # ---- common subexpressions ---------------------------------------------------------------------------------
  # `(x - m) * (x - m)` is two SUB nodes: emitted once, its register kept until the last use (one instruction less
  # per repeated subtree; the compiled tiers fold them anyway, the interpreter pays per instruction).
  def _key(self, v):
    k = self._key_of_node.get(id(v))
    if k is not None:
      return k
    if v.kind == 'tensor':
      t = ('t', self.input_reg(v))
    elif v.kind == 'const':
      val = v.value
      t = ('c', int(val)) if self.cls == _hip.SP_I64 else ('c', float(val).hex())
    elif v.kind == 'iota':
      t = ('i',)
    elif v.kind == 'op' and v.op == 'FILL':
      k = self._key(v.args[0])
      self._key_of_node[id(v)] = k
      return k
    elif v.kind == 'op':
      t = (v.op, v.dtype, v.args[0].dtype if v.op == 'CAST' else None) + tuple(self._key(a) for a in v.args)
    else:
      t = ('?', id(v))
    k = self._key_ids.setdefault(t, len(self._key_ids))
    self._key_of_node[id(v)] = k
    return k

  def _operands(self, v):
    """The children emit() evaluates into registers for `v`."""
    if v.kind != 'op':
      return []
    if v.op in _WITH_CONST and len(v.args) == 2:
      lhs, rhs = v.args
      if rhs.kind == 'const' and lhs.kind != 'const':
        return [lhs]
      if lhs.kind == 'const' and rhs.kind != 'const':
        return [rhs]
    return v.args

  def _count_uses(self, v):
    while v.kind == 'op' and v.op == 'FILL':
      v = v.args[0]
    k = self._key(v)
    n = self._uses.get(k, 0)
    self._uses[k] = n + 1
    if n == 0:
      for a in self._operands(v):
        self._count_uses(a)

  def input_reg(self, v):
    for i, t in enumerate(self.tensors):
      if self.in_vals[i][0].shape == v.shape and _same_storage(t, v.tensor):
        return i
    t, elem_strides = v.tensor, None
    if hasattr(t, 'is_contiguous') and not t.is_contiguous():
      # a strided view (slice / transpose of a tile): read in place through its own strides when the
      # program can express them, otherwise through one dense copy
      st = tuple(int(x) for x in t.stride())
      # (inner stride 1 only: a transposed view is better served by the LDS-tiled transposing copy)
      if tuple(t.shape) == v.shape and all(x >= 0 for x in st) and (not st or st[-1] in (0, 1)) \
          and self.contiguous is not None:
        elem_strides = st
      elif self.contiguous is not None:
        t = self.contiguous(t)
        v.tensor = t
    self.tensors.append(t)
    self.in_vals.append((v, broadcast_strides(v.shape, self.out_shape, elem_strides)))
    if len(self.tensors) > min(_hip.SP_MAX_INPUTS, _hip.SP_NREG - 1):
      raise ProgramTooLarge('too many tensor operands')
    return len(self.tensors) - 1

  def collect_inputs(self, v):
    if v.kind == 'tensor':
      self.input_reg(v)
    for a in v.args:
      self.collect_inputs(a)

  def alloc(self):
    if self.free:
      return self.free.pop()
    if self.next_temp is None:
      self.next_temp = len(self.tensors)
    if self.next_temp >= _hip.SP_NREG:
      raise ProgramTooLarge('expression needs more than %d registers' % _hip.SP_NREG)
    r = self.next_temp
    self.next_temp += 1
    return r

  def release(self, r):
    k = self._key_of_reg.get(r)
    if k is not None:
      self._uses[k] -= 1
      if self._uses[k] > 0:
        return                                 # another consumer of the same value is still to come
      del self._key_of_reg[r]
      del self._reg_of_key[k]
    if r >= len(self.tensors) and r not in self.free:
      self.free.append(r)

  def const_index(self, v):
    val = v.value
    if self.cls == _hip.SP_I64:
      val = int(val)
    elif self.cls == _hip.SP_F32 and not v.weak and v.dtype == np.float64:
      raise AssertionError('float64 constant in a float32 program')
    return self.prog.add_const(float(val) if self.cls != _hip.SP_I64 else int(val))

  def emit(self, v):
    if v.kind == 'tensor':
      return self.input_reg(v)
    if not self.share or (v.kind == 'op' and v.op == 'FILL'):
      return self._emit(v)
    k = self._key(v)
    k = self._same_as.get(k, k)
    r = self._reg_of_key.get(k)
    if r is None:
      r = self._emit(v)
      if r >= len(self.tensors):
        other = self._key_of_reg.get(r)
        if other is None:
          self._reg_of_key[k] = r
          self._key_of_reg[r] = k
        elif other != k:
          # (a cast that changes nothing returned its operand's register: this node is one more name for that value,
          #  and all its consumers release that register)
          self._same_as[k] = other
          self._uses[other] += self._uses.get(k, 1) - 1
    return r

  def _emit(self, v):
    p = self.prog
    if v.kind == 'tensor':
      return self.input_reg(v)
    if v.kind == 'const':
      r = self.alloc()
      p.emit('CONST', r, self.const_index(v))
      return r
    if v.kind == 'iota':
      r = self.alloc()
      p.emit('IOTA', r)
      return r
    if v.kind != 'op':
      raise NotLowerable('value of kind %r cannot be computed in a kernel' % v.kind)
    if v.op == 'FILL':
      return self.emit(v.args[0])
    if v.op == 'CAST':
      r = self.emit(v.args[0])
      src_dt, dst_dt = v.args[0].dtype, v.dtype
      op = None
      if dst_dt == np.bool_:
        op = None if src_dt == np.bool_ else 'TO_BOOL'
      elif dst_dt.kind in 'iu':
        name = {1: 'TO_U8', 4: 'TO_I32', 8: 'TO_I64'}[dst_dt.itemsize]
        if src_dt.kind == 'f':
          op = name                                   # truncate toward zero
        elif src_dt.kind in 'iu' and src_dt.itemsize > dst_dt.itemsize:
          op = name                                   # wrap
      elif dst_dt == np.float32:
        op = 'TO_F32' if self.cls != _hip.SP_F32 else None
      if op is None:
        return r
      self.release(r)
      dst = self.alloc()                      # (r itself when this was its last use)
      p.emit(op, dst, r)
      return dst
    # an operator with ONE constant operand is one instruction (`x + c` -> ADDC): no CONST into a register first
    with_const = None
    if v.op in _WITH_CONST and len(v.args) == 2:
      lhs, rhs = v.args
      if rhs.kind == 'const' and lhs.kind != 'const':
        with_const = (_WITH_CONST[v.op][0], lhs, rhs)
      elif lhs.kind == 'const' and rhs.kind != 'const':
        with_const = (_WITH_CONST[v.op][1], rhs, lhs)
    if with_const is not None:
      name, operand, const = with_const
      regs = [self.emit(operand)]
      self.release(regs[0])
      dst = self.alloc()
      p.emit(name, dst, self.const_index(const), regs[0])
    else:
      regs = [self.emit(a) for a in v.args]
      for r in regs:
        self.release(r)
      dst = self.alloc()
    if with_const is not None:
      pass
    elif v.op == 'WHERE':
      p.emit('WHERE', dst, regs[0], regs[1], regs[2])
    elif len(regs) == 1:
      p.emit(v.op, dst, regs[0])
    else:
      p.emit(v.op, dst, regs[0], regs[1])
    # keep the value inside its NumPy dtype when the class is wider
    if v.op not in _NO_NORMALISE_OPS and v.dtype in _NORMALISE and class_of(v.dtype) != self.cls:
      p.emit(_NORMALISE[v.dtype], dst, dst)
    elif v.op not in _NO_NORMALISE_OPS and v.dtype == np.int32 and self.cls == _hip.SP_I64:
      p.emit('TO_I32', dst, dst)
    elif v.op not in _NO_NORMALISE_OPS and v.dtype == np.uint8:
      p.emit('TO_U8', dst, dst)
    return dst

  def finish(self, root, out_dtype):
    self.collect_inputs(root)   # inputs first: they own registers 0..n-1
    self.next_temp = len(self.tensors)
    try:
      self.share = True
      self._count_uses(root)
      result = self.emit(root)
    except ProgramTooLarge:
      # a shared value holds its register until its last use: when that is what does not fit, evaluate every
      # occurrence on its own, as the tree says
      self.share = False
      self.prog = Program()
      self.free = []
      self.next_temp = len(self.tensors)
      self._reg_of_key.clear()
      self._key_of_reg.clear()
      result = self.emit(root)
    self.prog.result_reg = result
    self.prog.inputs = []
    strides = [st for (_, st) in self.in_vals]
    cshape, cstrides = collapse(self.out_shape, strides) if strides else collapse(self.out_shape, [])
    dense = dense_strides(cshape)
    linear = all(tuple(st) == dense or all(s == 0 for s in st) for st in cstrides)
    for (v, _), st in zip(self.in_vals, cstrides):
      self.prog.add_input(v.dtype, st)
    return self.prog.finish(self.cls, cshape, out_dtype, linear), self.tensors
