"""Process-group plumbing for one-process-per-GPU runs (launched by torch.distributed.run).

Control plane = torch.distributed on gloo (rendezvous, barrier, scalar reductions, broadcast of the
128-byte RCCL unique id); data plane = RCCL inside libsimgan_hip.so (sg_ctx_comm_init), issued on
the library's own stream.  Nothing here touches torch.cuda.
"""
import os


class ProcessGroup(object):
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if not dist.is_initialized():
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def broadcast_bytes(self, payload, src=0):
        """payload: bytes on `src`, ignored elsewhere -> the same bytes on every rank."""
        if self.dist is None:
            return payload
        box = [payload if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def max(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather(self, value):
        """One float per rank -> the list of all ranks' values (on every rank)."""
        if self.dist is None:
            return [float(value)]
        box = [None] * self.world
        self.dist.all_gather_object(box, float(value))
        return box

    def gather_object(self, obj):
        """Any picklable object per rank -> the list of all ranks' objects (on every rank)."""
        if self.dist is None:
            return [obj]
        box = [None] * self.world
        self.dist.all_gather_object(box, obj)
        return box

    def init_device_comm(self, ctx, make_unique_id):
        """Create the RCCL communicator on `ctx`: rank 0 draws the id, everyone joins."""
        if self.world == 1:
            return
        uid = self.broadcast_bytes(make_unique_id() if self.rank == 0 else None, src=0)
        ctx.comm_init(uid, self.rank, self.world)

    def shutdown(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
