"""The instruction streams lower.Emitter produces, run by a register machine written in NumPy (no GPU): random operator
trees -- with repeated subtrees, by identity and by structure, constants on either side, casts that change nothing --
must compute what NumPy computes for the tree.  Pins the emitter's register allocation: shared values keep their
register until their LAST use (round 5: common subexpressions are emitted once), operators with a constant operand
are one instruction, and a program that does not fit with shared values falls back to the tree as written."""
import numpy as np
import pytest

from spartan_amd import _hip, lower
from spartan_amd.program import ProgramTooLarge

NAMES = {v: k for k, v in _hip.OP.items()}


def run_stream(prog, tensors, dtype):
  """sp_interp.hpp's sp_step for the operators used below, on whole arrays."""
  n = prog.n_inputs
  regs = {i: np.asarray(tensors[i]).astype(dtype) for i in range(n)}
  const = (lambda i: dtype(prog.iconsts[i])) if np.dtype(dtype).kind == 'i' else (lambda i: dtype(prog.consts[i]))
  nanmax = lambda a, b: np.where(np.isnan(a), a, np.where(np.isnan(b), b, np.maximum(a, b)))
  nanmin = lambda a, b: np.where(np.isnan(a), a, np.where(np.isnan(b), b, np.minimum(a, b)))
  if np.dtype(dtype).kind == 'i':
    nanmax, nanmin = np.maximum, np.minimum
  with np.errstate(all='ignore'):
    for k in range(prog.n_instr):
      I = prog.instr[k]
      op = NAMES[I.op]
      if op == 'NOP':
        continue
      if op == 'CONST':
        regs[I.dst] = np.full(np.shape(regs[0]), const(I.a), dtype)
        continue
      if op.endswith('C') and I.op >= 60:
        assert I.b in regs, 'instruction %d reads register %d that nothing has written' % (k, I.b)
        x, c = regs[I.b], const(I.a)
        regs[I.dst] = {'ADDC': lambda: x + c, 'SUBC': lambda: x - c, 'RSUBC': lambda: c - x, 'MULC': lambda: x * c,
                       'DIVC': lambda: x / c, 'RDIVC': lambda: c / x, 'MAXC': lambda: nanmax(x, c),
                       'MINC': lambda: nanmin(x, c)}[op]().astype(dtype)
        continue
      assert I.a in regs, 'instruction %d reads register %d that nothing has written' % (k, I.a)
      a = regs[I.a]
      binary = 10 <= I.op <= 28
      if binary or op == 'WHERE':
        assert I.b in regs, 'instruction %d reads register %d that nothing has written' % (k, I.b)
      b = regs.get(I.b)
      fn = {'ADD': lambda: a + b, 'SUB': lambda: a - b, 'MUL': lambda: a * b, 'DIV': lambda: a / b,
            'MAX': lambda: nanmax(a, b), 'MIN': lambda: nanmin(a, b), 'GT': lambda: a > b, 'LT': lambda: a < b,
            'NEG': lambda: -a, 'ABS': lambda: np.abs(a), 'SQRT': lambda: np.sqrt(a), 'SQUARE': lambda: a * a,
            'TO_BOOL': lambda: a != 0, 'TO_F32': lambda: a.astype(np.float32),
            'WHERE': lambda: np.where(a != 0, b, regs[I.c]), 'MOV': lambda: a}[op]
      regs[I.dst] = np.asarray(fn()).astype(dtype)
  return regs[prog.result_reg]


class Tree(object):
  """Grows a random tree as (V, NumPy value) pairs; `pool` feeds earlier subtrees back in, as the same object or as
  a structurally equal copy."""

  def __init__(self, rng, xs, dtype):
    self.rng, self.dtype = rng, np.dtype(dtype)
    self.leaves = [(lambda x=x: lower.V('tensor', dtype=self.dtype, shape=x.shape, tensor=x), x) for x in xs]
    self.pool = []

  def leaf(self):
    mk, x = self.leaves[self.rng.randint(len(self.leaves))]
    return mk(), x, ('leaf', id(x))

  def const(self):
    c = float(self.rng.randint(1, 6)) / 2 if self.dtype.kind == 'f' else int(self.rng.randint(1, 6))
    return lower.const(c), c, ('const', c)

  def rebuild(self, recipe):
    """A fresh V tree (new node objects) from a recipe."""
    if recipe[0] == 'leaf':
      for mk, x in self.leaves:
        if id(x) == recipe[1]:
          return mk(), x
    if recipe[0] == 'const':
      return lower.const(recipe[1]), recipe[1]
    name, fn = recipe[0], recipe[1]
    parts = [self.rebuild(r) for r in recipe[2:]]
    with np.errstate(all='ignore'):
      vals = [np.asarray(p[1], self.dtype) if not np.isscalar(p[1]) else self.dtype.type(p[1]) for p in parts]
      if name == 'WHERE':
        vals[0] = np.asarray(parts[0][1]) != 0
      val = fn(*vals)
    return lower.apply(name, fn, [p[0] for p in parts]), val

  def grow(self, depth):
    r = self.rng.rand()
    if self.pool and r < 0.3:
      v, val, recipe = self.pool[self.rng.randint(len(self.pool))]
      if self.rng.rand() < 0.5:
        return v, val, recipe                          # the same node object again
      v2, val2 = self.rebuild(recipe)                  # an equal subtree made of new nodes
      return v2, val2, recipe
    if depth == 0 or r < 0.4:
      return self.leaf()
    binary = [('ADD', np.add), ('SUB', np.subtract), ('MUL', np.multiply), ('MAX', np.maximum), ('MIN', np.minimum)]
    if self.dtype.kind == 'f':
      binary.append(('DIV', np.divide))
    unary = [('NEG', np.negative), ('ABS', np.abs)]
    if self.rng.rand() < 0.15:
      # where(a > b, c, d): the comparison lives inside this node only (booleans do not subtract or negate in NumPy)
      a, b = self.grow(depth - 1), self.grow(depth - 1)
      c = self.const() if self.rng.rand() < 0.3 else self.grow(depth - 1)
      d = self.grow(depth - 1)
      cmp_name, cmp_fn = [('GT', np.greater), ('LT', np.less)][self.rng.randint(2)]
      if a[2][0] == 'const' and b[2][0] == 'const':
        a = self.leaf()
      cond = lower.apply(cmp_name, cmp_fn, [a[0], b[0]])
      with np.errstate(all='ignore'):
        cval = cmp_fn(np.asarray(a[1], self.dtype), np.asarray(b[1], self.dtype))
        val = np.where(cval, np.asarray(c[1], self.dtype), np.asarray(d[1], self.dtype))
      node = (lower.apply('WHERE', np.where, [cond, c[0], d[0]]), np.asarray(val, self.dtype),
              ('WHERE', np.where, (cmp_name, cmp_fn, a[2], b[2]), c[2], d[2]))
      self.pool.append(node)
      return node
    if self.rng.rand() < 0.2:
      name, fn = unary[self.rng.randint(len(unary))]
      a = self.grow(depth - 1)
      parts = [a]
    else:
      name, fn = binary[self.rng.randint(len(binary))]
      a = self.grow(depth - 1)
      b = self.const() if self.rng.rand() < 0.35 else self.grow(depth - 1)
      parts = [a, b] if self.rng.rand() < 0.5 else [b, a]
      if all(p[2][0] == 'const' for p in parts):
        parts[0] = self.leaf()
    with np.errstate(all='ignore'):
      val = fn(*[np.asarray(p[1], self.dtype) if not np.isscalar(p[1]) else self.dtype.type(p[1]) for p in parts])
    node = (lower.apply(name, fn, [p[0] for p in parts]), np.asarray(val, self.dtype), (name, fn) + tuple(p[2] for p in parts))
    self.pool.append(node)
    return node


@pytest.mark.parametrize('dtype', [np.float32, np.int64])
def test_random_trees_with_repeated_subtrees(dtype):
  rng = np.random.RandomState(5)
  cls = lower.class_of(np.dtype(dtype))
  fitted = shared = 0
  for trial in range(400):
    xs = [(rng.rand(6, 10) * 8 - 4).astype(dtype) for _ in range(rng.randint(1, 4))]
    if np.dtype(dtype).kind == 'i':
      xs = [np.where(x == 0, 3, x) for x in xs]
    t = Tree(rng, xs, dtype)
    v, want, _ = t.grow(rng.randint(2, 6))
    if v.kind != 'op':
      continue
    em = lower.Emitter(cls, v.shape)
    try:
      prog, tensors = em.finish(v, np.dtype(dtype))
    except ProgramTooLarge:
      continue
    fitted += 1
    shared += em.share
    got = run_stream(prog, tensors, dtype)
    np.testing.assert_array_equal(got, np.asarray(want, dtype), err_msg='trial %d' % trial)
    ops = [NAMES[prog.instr[i].op] for i in range(prog.n_instr)]
    assert 'CONST' not in ops or 'WHERE' in ops, ops       # (only where() takes a constant through a register)
  assert fitted > 150 and shared > 100, (fitted, shared)


def test_a_repeated_subtree_is_one_instruction():
  x = np.arange(12, dtype=np.float32).reshape(3, 4)
  T = lambda: lower.V('tensor', dtype=np.dtype(np.float32), shape=x.shape, tensor=x)
  dev = lambda: lower.apply('SUB', np.subtract, [T(), lower.const(0.5)])
  root = lower.apply('MUL', np.multiply, [dev(), dev()])
  prog, tensors = lower.Emitter(_hip.SP_F32, root.shape).finish(root, np.dtype(np.float32))
  assert [NAMES[prog.instr[i].op] for i in range(prog.n_instr)] == ['SUBC', 'MUL']
  np.testing.assert_array_equal(run_stream(prog, tensors, np.float32), (x - np.float32(0.5)) ** 2)


def test_a_shared_value_outlives_the_registers_between_its_uses():
  """d = x - y is used first and last; five temporaries come and go in between."""
  rng = np.random.RandomState(2)
  x, y = rng.rand(4, 4).astype(np.float32), rng.rand(4, 4).astype(np.float32)
  X = lambda: lower.V('tensor', dtype=np.dtype(np.float32), shape=x.shape, tensor=x)
  Y = lambda: lower.V('tensor', dtype=np.dtype(np.float32), shape=y.shape, tensor=y)
  ap = lower.apply
  d = ap('SUB', np.subtract, [X(), Y()])
  acc, want = d, x - y
  for k in range(5):
    acc = ap('ADD', np.add, [ap('MUL', np.multiply, [acc, Y()]), ap('MUL', np.multiply, [X(), X()])])
    want = want * y + x * x
  root = ap('MUL', np.multiply, [acc, ap('SUB', np.subtract, [X(), Y()])])
  prog, tensors = lower.Emitter(_hip.SP_F32, root.shape).finish(root, np.dtype(np.float32))
  ops = [NAMES[prog.instr[i].op] for i in range(prog.n_instr)]
  assert ops.count('SUB') == 1 and ops.count('MUL') == 7, ops        # x * x once, too
  np.testing.assert_array_equal(run_stream(prog, tensors, np.float32), want * (x - y))


def test_a_cast_that_changes_nothing_shares_its_operands_register():
  x = np.arange(8, dtype=np.float32)
  X = lambda: lower.V('tensor', dtype=np.dtype(np.float32), shape=x.shape, tensor=x)
  ap = lower.apply
  sq = ap('MUL', np.multiply, [X(), X()])
  same = lower.V('op', dtype=np.float32, shape=x.shape, op='CAST', args=[sq])       # float32 -> float32
  root = ap('ADD', np.add, [ap('ADD', np.add, [same, same]), ap('MUL', np.multiply, [X(), X()])])
  em = lower.Emitter(_hip.SP_F32, root.shape)
  prog, tensors = em.finish(root, np.dtype(np.float32))
  ops = [NAMES[prog.instr[i].op] for i in range(prog.n_instr)]
  assert ops == ['MUL', 'ADD', 'ADD'], ops
  np.testing.assert_array_equal(run_stream(prog, tensors, np.float32), 3 * x * x)
  assert not em._key_of_reg or set(em._key_of_reg) == {prog.result_reg}              # every other register was returned


def test_values_shared_beyond_the_register_file_fall_back_to_the_tree_as_written():
  """Seven products, each needed again after all the others: kept, they would hold seven registers while the sum
  needs an eighth beside the operand's; evaluated where they stand, two registers do."""
  x = np.linspace(-2, 2, 24, dtype=np.float32).reshape(4, 6)
  X = lambda: lower.V('tensor', dtype=np.dtype(np.float32), shape=x.shape, tensor=x)
  ap = lower.apply
  f = lambda k: ap('MUL', np.multiply, [X(), lower.const(0.25 * (k + 1))])
  root, want = f(0), x * np.float32(0.25)
  for k in list(range(1, 7)) + list(range(7)):
    root = ap('ADD', np.add, [root, f(k)])
    want = want + x * np.float32(0.25 * (k + 1))
  em = lower.Emitter(_hip.SP_F32, root.shape)
  prog, tensors = em.finish(root, np.dtype(np.float32))
  assert not em.share
  ops = [NAMES[prog.instr[i].op] for i in range(prog.n_instr)]
  assert ops.count('MULC') == 14 and ops.count('ADD') == 13, ops
  np.testing.assert_array_equal(run_stream(prog, tensors, np.float32), want)
  # one product fewer fits: six kept values, the running sum, the operand
  root = f(0)
  for k in list(range(1, 6)) + list(range(6)):
    root = ap('ADD', np.add, [root, f(k)])
  em = lower.Emitter(_hip.SP_F32, root.shape)
  prog, tensors = em.finish(root, np.dtype(np.float32))
  assert em.share and [NAMES[prog.instr[i].op] for i in range(prog.n_instr)].count('MULC') == 6


def test_streams_of_the_prebuilt_kernel_library_are_recognised_without_a_gpu():
  """sp_program_static_id is host code: the emitter's streams for the library's expressions map to their ids (the GPU
  twin, test_hip_kernels.py::test_static_program_library, also runs them)."""
  import ctypes as C
  x, y = np.zeros((64, 128), np.float32), np.ones((64, 128), np.float32)
  yp, yy = np.zeros((64, 1), np.float32), np.ones((64, 1), np.float32)
  T = lambda t: lower.V('tensor', dtype=np.float32, shape=tuple(t.shape), tensor=t)
  ap = lower.apply
  cases = {
      1: ap('ADD', np.add, [T(x), lower.const(1)]), 2: ap('SUB', np.subtract, [T(x), lower.const(1.5)]),
      3: ap('MUL', np.multiply, [T(x), lower.const(2.0)]), 4: ap('DIV', np.divide, [T(x), lower.const(7)]),
      5: ap('ADD', np.add, [T(x), T(y)]), 6: ap('SUB', np.subtract, [T(x), T(y)]),
      7: ap('MUL', np.multiply, [T(x), T(y)]), 8: ap('DIV', np.divide, [T(x), T(y)]),
      9: ap('ADD', np.add, [ap('MUL', np.multiply, [T(x), T(x)]), T(x)]),
      10: ap('MUL', np.multiply, [T(x), ap('SUB', np.subtract, [T(yp), T(yy)])]),
      11: ap('MUL', np.multiply, [T(x), T(x)]), 12: ap('SUB', np.subtract, [lower.const(3.0), T(x)])}
  for sid, root in cases.items():
    prog, _ = lower.Emitter(_hip.SP_F32, root.shape).finish(root, np.float32)
    assert _hip.lib().sp_program_static_id(C.byref(prog), _hip.SP_F32) == sid, sid
  ident = lower.Emitter(_hip.SP_F32, x.shape).finish(T(x), None)[0]
  assert _hip.lib().sp_program_static_id(C.byref(ident), -1) == 0
