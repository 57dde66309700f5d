"""One rank over the driver's backend ("nccl" = RCCL): the exchange primitive of a sharded run (ShardTransport.allreduce:
in-place all-reduce on a VIEW of the int32 exchange tensor, SUM and MAX) and bench.py's timing collectives (barrier +
MAX over a float64 CUDA tensor), against the real library.  A second rank needs a second GPU (tests/test_gpu_procs.py
test_two_ranks_over_rccl); what one rank can show is that the calls the ranks make exist, take these dtypes and ops,
and leave the data where the library expects it."""
import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    fdist = importlib.import_module("founder-sequences_amd.dist")
    tr = fdist.ShardTransport(1 << 16, torch.device("cuda", 0), dist)
    ref = torch.arange(1 << 16, dtype=torch.int32, device="cuda")
    tr.buf.copy_(ref)
    assert tr.allreduce(0, 1 << 16, 0) == 0                      # SUM over one rank: unchanged
    assert tr.allreduce(12345, 777, 1) == 0                      # MAX on a view in the middle
    assert tr.allreduce((1 << 16) - 1, 1, 1) == 0                # the status word (last word of the buffer)
    assert tr.allreduce(5, 0, 0) == 0                            # nothing to exchange
    assert torch.equal(tr.buf, ref) and tr.calls == 4
    # words >= 2^31 travel as negative int32: a SUM of one contribution is still exact
    tr.buf[:4] = torch.tensor([-1, -2147483648, 2147483647, 0], dtype=torch.int32, device="cuda")
    keep = tr.buf[:4].clone()
    assert tr.allreduce(0, 4, 0) == 0 and torch.equal(tr.buf[:4], keep)
    # bench.py's timed region: barrier, synchronize, MAX over ranks of the elapsed time
    ms = fdist.timed_steps(lambda: None, 3, 1, dist=dist, device_sync=torch.cuda.synchronize,
                           tensor_factory=lambda v: torch.tensor(v, dtype=torch.float64, device="cuda"))
    assert ms >= 0.0
    dist.destroy_process_group()
    print("rccl transport ok")


if __name__ == "__main__":
    main()
