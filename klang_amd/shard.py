"""Multi-GPU sharding of a synth bank: one process per GPU, voices partitioned by contiguous
ranges of synth instances, no data-path collective except ONE all-reduce (RCCL over xGMI; `nccl` backend on ROCm)
of the [2][n] stereo block per step, and only when the bank spans more than one GPU.

The renderer is injectable (`bank_factory`) so the partition / event-routing / reduce logic can be exercised on CPU
with the gloo backend (tests/test_sharding_gloo.py); the product default is the HIP SynthBank.
"""
from .bank import FxBank, SynthBank


def shard_range(n_items, world, rank):
    """Contiguous split of n_items over `world` ranks; the first n_items % world ranks own one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def owner_of(item, n_items, world):
    base, extra = divmod(n_items, world)
    split = extra * (base + 1)
    return item // (base + 1) if item < split else extra + (item - split) // max(base, 1)


class ShardedSynthBank:
    def __init__(self, patch, synths, notes, fs=48000.0, max_block=256, rank=0, world=1, device=None,
                 bank_factory=None, all_reduce=None):
        self.rank, self.world, self.synths, self.notes = rank, world, synths, notes
        self.lo, self.hi = shard_range(synths, world, rank)
        if self.hi <= self.lo:
            raise ValueError(f"rank {rank} of {world} owns no synth instance (synths={synths})")
        factory = bank_factory or (lambda **kw: SynthBank(device=device, **kw))
        self.bank = factory(patch=patch, synths=self.hi - self.lo, notes=notes, fs=fs, max_block=max_block)
        self._all_reduce = all_reduce
        self.voices = self.bank.voices
        self.total_voices = synths * notes

    def owns(self, synth):
        return self.lo <= synth < self.hi

    # events: every rank sees the same (global) event stream and keeps its own share
    def random(self, seed):
        self.bank.random(seed)

    def note_on(self, synth, pitch, velocity=1.0, seed=None):
        if not self.owns(synth):
            return None
        if seed is not None:
            self.bank.random(seed)
        return self.bank.note_on(synth - self.lo, pitch, velocity)

    def note_on_many(self, synths, pitches, velocities):
        import numpy as np
        sy = np.asarray(synths); m = (sy >= self.lo) & (sy < self.hi)
        return self.bank.note_on_many(sy[m] - self.lo, np.asarray(pitches)[m], np.asarray(velocities)[m])

    def note_off_many(self, synths, pitches, velocities):
        import numpy as np
        sy = np.asarray(synths); m = (sy >= self.lo) & (sy < self.hi)
        return self.bank.note_off_many(sy[m] - self.lo, np.asarray(pitches)[m], np.asarray(velocities)[m])

    def note_off(self, synth, pitch, velocity=0.0):
        if self.owns(synth):
            self.bank.note_off(synth - self.lo, pitch, velocity)

    def set_control(self, synth, index, value):
        if self.owns(synth):
            self.bank.set_control(synth - self.lo, index, value)

    def process_device(self, mix, n, stream=None, async_reduce=False):
        """mix: torch tensor [2][n] on this rank's device, ACCUMULATED into; afterwards every rank holds the global sum.
        async_reduce=True (throughput mode): the all-reduce is only enqueued and its work handle returned — call .wait()
        on it before touching `mix` again; the next block's render then overlaps the (latency-bound, 2 KiB) collective."""
        self.bank.process_device(mix.data_ptr(), n, stream)
        if self.world > 1:
            if self._all_reduce is not None:
                self._all_reduce(mix)
            else:
                import torch.distributed as dist
                if async_reduce:
                    return dist.all_reduce(mix, async_op=True)
                dist.all_reduce(mix)
        return None

    def close(self):
        self.bank.close()


class ShardedFxBank:
    """SURVEY.md §8e for effect banks: instances are independent (one Stereo::Effect object each in the reference), so they shard BY INSTANCE over the
    ranks — contiguous ranges, like the synths above — and there is NO collective on the data path: a rank processes the rows of the caller's block that
    belong to its instances and nobody else's.  Dials go to the rank that owns the instance.  (Inside one process the library shards a bank over the GPUs
    of klg_init itself; this is the one-process-per-GPU form.)"""

    def __init__(self, patch, instances, fs=48000.0, max_block=256, rank=0, world=1, device=None, bank_factory=None, **kw):
        self.rank, self.world, self.instances = rank, world, instances
        self.lo, self.hi = shard_range(instances, world, rank)
        if self.hi <= self.lo:
            raise ValueError(f"rank {rank} of {world} owns no effect instance (instances={instances})")
        factory = bank_factory or (lambda *a, **k: FxBank(*a, device=device, **k))
        self.bank = factory(patch, self.hi - self.lo, fs=fs, max_block=max_block, **kw)

    def owns(self, instance):
        return self.lo <= instance < self.hi

    def set_control(self, instance, index, value):
        if self.owns(instance):
            self.bank.set_control(instance - self.lo, index, value)

    def get_control(self, instance, index):
        """The control as the owning rank's effect left it; None on the other ranks."""
        return self.bank.get_control(instance - self.lo, index) if self.owns(instance) else None

    def local(self, io):
        """This rank's rows of a global [instances][channels][n] block (a view: processing it in place processes the caller's block)."""
        return io[self.lo:self.hi]

    def process(self, local_io):
        """local_io: float32 [hi - lo][channels][n] — this rank's instances, processed in place (host buffers, synchronous)."""
        return self.bank.process(local_io)

    def process_device(self, d_local_io_ptr, n, stream=None):
        return self.bank.process_device(d_local_io_ptr, n, stream)

    def close(self):
        self.bank.close()
