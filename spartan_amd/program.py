"""Builder for `sp_program`: the serialised form of a fused LocalExpr tree.

The reference evaluates a fused tree node by node with NumPy
(spartan/expr/operator/local.py:115-127), materialising one full-tile temporary
per node.  Here the tree is flattened into a straight-line register program that
the HIP kernels evaluate per element (spartan_amd/csrc/sp_interp.hpp).
"""
import ctypes as C

import numpy as np

from . import _hip
from ._hip import OP, SP_F32, SP_F64, SP_I64, SP_MAX_CONSTS, SP_MAX_DIMS, SP_MAX_INPUTS, SP_MAX_INSTR, SP_NREG


class ProgramTooLarge(Exception):
  """The tree does not fit one kernel (inputs / registers / instructions)."""


def class_of(dtype):
  """Arithmetic class (sp_program.cls) able to represent `dtype` exactly."""
  dtype = np.dtype(dtype)
  if dtype == np.float32:
    return SP_F32
  if dtype == np.float64:
    return SP_F64
  if dtype.kind in 'iub':
    return SP_I64
  raise TypeError('unsupported dtype %s' % dtype)


def join_class(a, b):
  """Smallest class containing both (float32 (+) int64 -> float64, as NumPy promotes)."""
  if a == b:
    return a
  s = {a, b}
  if SP_F64 in s:
    return SP_F64
  # {F32, I64}
  return SP_F64


class Program(object):
  """Mutable builder; `finish()` returns the ctypes struct."""

  def __init__(self):
    self.inputs = []      # list of (dtype_code, strides tuple)
    self.instrs = []      # (op, dst, a, b, c)
    self.consts = []      # python numbers
    self.result_reg = 0

  def add_input(self, np_dtype, strides):
    if len(self.inputs) >= min(SP_MAX_INPUTS, SP_NREG):
      raise ProgramTooLarge('more than %d inputs' % min(SP_MAX_INPUTS, SP_NREG))
    self.inputs.append((_hip.sp_dtype(np_dtype), tuple(int(s) for s in strides)))
    return len(self.inputs) - 1

  def add_const(self, value):
    for i, c in enumerate(self.consts):
      if c == value and type(c) == type(value) and not (isinstance(value, float) and value != value):
        return i
    if len(self.consts) >= SP_MAX_CONSTS:
      raise ProgramTooLarge('more than %d constants' % SP_MAX_CONSTS)
    self.consts.append(value)
    return len(self.consts) - 1

  def emit(self, op, dst, a=0, b=0, c=0):
    if len(self.instrs) >= SP_MAX_INSTR:
      raise ProgramTooLarge('more than %d instructions' % SP_MAX_INSTR)
    self.instrs.append((OP[op] if isinstance(op, str) else int(op), dst, a, b, c))

  def finish(self, cls, shape, out_dtype, linear):
    shape = tuple(int(s) for s in shape)
    if len(shape) == 0:
      shape = (1,)
    if len(shape) > SP_MAX_DIMS:
      raise ProgramTooLarge('index space has %d dims (max %d)' % (len(shape), SP_MAX_DIMS))
    p = _hip.sp_program()
    p.cls = cls
    p.n_inputs = len(self.inputs)
    p.n_instr = len(self.instrs)
    p.result_reg = self.result_reg
    p.ndim = len(shape)
    p.out_dtype = _hip.sp_dtype(out_dtype) if out_dtype is not None else 0
    p.linear = 1 if linear else 0
    for d, s in enumerate(shape):
      p.shape[d] = s
    for j, (dt, strides) in enumerate(self.inputs):
      p.in_dtype[j] = dt
      assert len(strides) == len(shape), (strides, shape)
      for d, s in enumerate(strides):
        p.in_stride[j][d] = s
    for i, cval in enumerate(self.consts):
      p.consts[i] = float(cval)
      try:
        p.iconsts[i] = int(cval)
      except (OverflowError, ValueError):
        p.iconsts[i] = 0
    for i, (op, dst, a, b, c) in enumerate(self.instrs):
      ins = p.instr[i]
      ins.op, ins.dst, ins.a, ins.b, ins.c = op, dst, a, b, c
    return p


def dense_strides(shape):
  """Row-major element strides of a dense array of `shape`."""
  out = []
  s = 1
  for n in reversed(shape):
    out.append(s)
    s *= int(n)
  return tuple(reversed(out))


def broadcast_strides(in_shape, out_shape, elem_strides=None):
  """Element strides of an array of `in_shape` viewed (NumPy broadcasting,
  right-aligned: reference spartan/expr/operator/broadcast.py:111-158) in
  `out_shape`; broadcast dimensions get stride 0.  `elem_strides`: the array's own
  element strides when it is a strided view (default: dense row-major)."""
  in_shape = tuple(in_shape)
  out_shape = tuple(out_shape)
  pad = len(out_shape) - len(in_shape)
  assert pad >= 0, (in_shape, out_shape)
  ds = dense_strides(in_shape) if elem_strides is None else tuple(int(s) for s in elem_strides)
  strides = [0] * pad
  for d, n in enumerate(in_shape):
    if n == out_shape[pad + d]:
      strides.append(ds[d] if n != 1 else 0)
    elif n == 1:
      strides.append(0)
    else:
      raise ValueError('cannot broadcast %s to %s' % (in_shape, out_shape))
  return tuple(strides)


def collapse(shape, strides_list):
  """Merge adjacent dimensions that are mergeable for EVERY operand (and for
  the dense output) so that the index space fits SP_MAX_DIMS.

  Returns (new_shape, new_strides_list).  Two adjacent dims (d, d+1) merge when
  for each operand stride[d] == stride[d+1] * shape[d+1] (dense run) or both
  strides are 0 (broadcast run).  Size-1 dims are dropped first.
  """
  shape = [int(s) for s in shape]
  strides_list = [list(s) for s in strides_list]
  # drop size-1 dims
  keep = [d for d, n in enumerate(shape) if n != 1]
  if not keep:
    return (1,), [(0,) for _ in strides_list]
  shape = [shape[d] for d in keep]
  strides_list = [[s[d] for d in keep] for s in strides_list]
  d = len(shape) - 2
  while d >= 0:
    ok = True
    for s in strides_list:
      if not ((s[d] == s[d + 1] * shape[d + 1]) or (s[d] == 0 and s[d + 1] == 0)):
        ok = False
        break
    if ok:
      n = shape[d] * shape[d + 1]
      shape[d:d + 2] = [n]
      for s in strides_list:
        s[d:d + 2] = [s[d + 1]]
    d -= 1
  return tuple(shape), [tuple(s) for s in strides_list]
