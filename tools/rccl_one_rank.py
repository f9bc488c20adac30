"""The RCCL data plane in a process that never imports torch: one-rank communicator through sp_comm_*, every
primitive self-tested, and the files it runs on (sp_comm_paths + /proc/self/maps).  python tools/rccl_one_rank.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import comm, _hip
assert comm.gpu_count() >= 1, 'needs a GPU'
_hip.check(_hip.lib().sp_set_device(0))
t = comm.RcclTransport(1, 0, comm.RcclTransport.unique_id())
ok, msg = t.self_test(30.0)
t.close()
print(json.dumps({'self_test': [ok, msg], 'paths': comm.rccl_paths(), 'mapped': comm.mapped_runtimes(),
                  'torch_in_process': 'torch' in sys.modules}, indent=1))
assert ok and 'torch' not in sys.modules
