"""
Utterance sharding over the GPUs of one node.

The reference parallelises this path with ``split_scp.pl`` + ``run.pl JOB=1:nj``
(scripts/run_adapt_beamformer.sh:69-92): contiguous scp shards, one process
each, no communication.  Here: one process per GPU (torchrun), utterances are
independent units dealt to ranks by duration (longest first, each to the least
loaded rank) and
RCCL (torch.distributed backend "nccl") carries only the start/finish barrier
and the three counters of the final "Processed N utterances" line.  There is no
data-path collective because the path has no exchange step.
"""
import os


class Shard:
    """rank/world view of the job; degenerates to a single process."""

    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self._dist = None
        self.assigned_weight = 0.0
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                # the ranks can only agree on a port through their launcher
                raise RuntimeError("WORLD_SIZE > 1 without MASTER_PORT: start the ranks with "
                                   "`python -m torch.distributed.run --master-addr 127.0.0.1 "
                                   "--master-port <free port> ...` (or export MASTER_PORT)")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if not dist.is_initialized():
                kw = {}
                if backend == "nccl":
                    torch.cuda.set_device(self.local_rank)
                    kw["device_id"] = torch.device("cuda", self.local_rank)
                dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self._dist = dist
            self.backend = backend

    @property
    def device(self):
        return self.local_rank

    def assign(self, keys, weights=None):
        """Keys owned by this rank.  With weights (e.g. durations) the keys are
        dealt longest first, each to the least loaded rank, which balances the sum
        of weights; without, plain round robin in table order."""
        return assign_keys(keys, self.rank, self.world, weights)

    def assign_by_duration(self, wav_reader, keys=None):
        """Keys of this rank, dealt longest-first / least-loaded on the sample counts
        in the wave headers (utterances whose length cannot be read from a header --
        pipes, per-channel globs -- count as the mean of the others)."""
        keys = list(wav_reader.index_keys if keys is None else keys)
        if self.world <= 1:
            return keys
        lens = [wav_reader.peek_nsamps(k) for k in keys]
        known = [n for n in lens if n]
        mean = (sum(known) / len(known)) if known else 1.0
        weights = [n if n else mean for n in lens]
        mine = assign_keys(keys, self.rank, self.world, weights)
        wsum = dict(zip(keys, weights))
        self.assigned_weight = sum(wsum[k] for k in mine)
        return mine

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def sum_counts(self, values):
        """Element-wise sum of a short list of python numbers over all ranks."""
        if self._dist is None:
            return list(values)
        import torch
        dev = torch.device("cuda", self.local_rank) if self.backend == "nccl" else "cpu"
        t = torch.tensor(list(values), dtype=torch.float64, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return t.cpu().tolist()

    def close(self):
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
            self._dist = None


def assign_keys(keys, rank, world, weights=None):
    keys = list(keys)
    if world <= 1:
        return keys
    if weights is None:
        return keys[rank::world]
    # longest first, each to the least loaded rank so far (LPT; ties -> lowest
    # rank): every rank evaluates the same deterministic deal
    order = sorted(range(len(keys)), key=lambda i: (-float(weights[i]), i))
    load = [0.0] * world
    owner = {}
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[i] = r
        load[r] += float(weights[i])
    mine = sorted(i for i, r in owner.items() if r == rank)  # table order inside a rank
    return [keys[i] for i in mine]
