"""One-shot all-reduce between the GPUs of one node (csrc/collective.hip; C ABI `yunet_comm_*` / `yunet_allreduce`).

What the reference does with torch DDP over NCCL and `reduce_mean` (mmdet/apis/train.py:152-163,
mmdet/core/utils/dist_utils.py:68-74) is, on this path, three latency-bound messages per step: num_pos (4 bytes),
the exposed gradient bucket (~50 KB) and the overlapped one (~250 KB).  xGMI is a point-to-point mesh, so every rank
stores its message straight into every peer's inbox and sums the world's slots in rank order -- one kernel per rank,
identical bits on every rank, no ring.  `torch.distributed` is used only to exchange the 64-byte IPC handles.

Opt-in (`YUNET_ONESHOT_AR=1`, `engine.enable_oneshot()`): it has been executed with two processes sharing ONE GPU
(tests/test_oneshot_gpu.py) -- the only multi-process configuration this build environment has -- and `verify()`
checks it against the process group's own all-gather before the engine relies on it.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L


class OneShotAllReduce:
    """One communicator = one inbox per rank + the peers' inboxes mapped into this process.  Use one communicator
    per stream that carries collectives: every rank must issue the same sequence of calls on it."""

    def __init__(self, device, max_bytes, group=None):
        if not dist.is_initialized():
            raise RuntimeError('OneShotAllReduce needs an initialised process group (handle exchange)')
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > L.MAX_RANKS:       # the same on every rank: raising here cannot strand a peer
            raise ValueError(f'one node only: world size {self.world} > {L.MAX_RANKS}')
        self.device = torch.device(device)
        self.lib = L.load()
        self._mapped = []
        self._inbox = self._status = None
        # Set-up has two local steps that can fail on ONE rank only (allocation / export of the inbox; mapping the
        # peers' handles: hipIpcOpenMemHandle denied in a container, a peer on another node).  A rank that raised
        # there alone would leave its peers blocked in the next collective, so every rank first reports how its
        # step went and all ranks raise TOGETHER; engine.enable_oneshot() turns that into "stay on the
        # process group" on every rank.
        err, handle_bytes, nbytes = None, None, 0
        try:
            with torch.cuda.device(self.device):
                nbytes = self.lib.yunet_comm_inbox_bytes(self.world, int(max_bytes))
                if nbytes == 0:
                    raise ValueError('bad world size / message size')
                inbox, status = C.c_void_p(), C.c_void_p()
                L.check(self.lib.yunet_comm_alloc(nbytes, C.byref(inbox), C.byref(status)), 'yunet_comm_alloc')
                self._inbox, self._status = inbox.value, status.value
                handle = C.create_string_buffer(L.IPC_HANDLE_BYTES)
                L.check(self.lib.yunet_comm_export(self._inbox, handle), 'yunet_comm_export (hipIpcGetMemHandle)')
                handle_bytes = bytes(handle.raw)
        except Exception as e:          # noqa: BLE001 -- reported to every rank below
            err = f'rank {self.rank}: {e!r}'
        reports = [None] * self.world
        dist.all_gather_object(reports, (err, handle_bytes, self.device.index if self.device.index is not None
                                         else torch.cuda.current_device()), group=group)
        self._agree([r[0] for r in reports])
        err = None
        try:
            # peer access between every pair of devices involved (one node): mapping a peer's inbox without it would turn
            # the first store into a memory fault instead of an error code
            mine = reports[self.rank][2]
            for r, (_, _, idx) in enumerate(reports):
                if idx != mine and not torch.cuda.can_device_access_peer(mine, idx):
                    raise RuntimeError(f'device {mine} cannot access its peer {idx} (rank {r}): no xGMI / PCIe peer path')
            with torch.cuda.device(self.device):
                comm = L.YunetComm()
                comm.rank, comm.world, comm.seq = self.rank, self.world, 0
                comm.slot_bytes = (nbytes - L.COMM_HEADER_BYTES) // (2 * self.world)
                comm.status = self._status
                for r, (_, h, _) in enumerate(reports):
                    if r == self.rank:
                        comm.inbox[r] = self._inbox
                        continue
                    mapped = C.c_void_p()
                    L.check(self.lib.yunet_comm_open(C.create_string_buffer(h, L.IPC_HANDLE_BYTES), C.byref(mapped)),
                            f'yunet_comm_open (hipIpcOpenMemHandle, rank {r})')
                    self._mapped.append(mapped.value)
                    comm.inbox[r] = mapped.value
                self.comm = comm
                self.max_bytes = int(comm.slot_bytes)
        except Exception as e:          # noqa: BLE001
            err = f'rank {self.rank}: {e!r}'
        opened = [None] * self.world
        dist.all_gather_object(opened, err, group=group)    # also the barrier: every inbox is mapped everywhere
        self._agree(opened)

    def _agree(self, errors):
        """All ranks hold the same list: raise on all of them if any rank failed its local step."""
        bad = [e for e in errors if e]
        if bad:
            self.close()
            raise RuntimeError('one-shot all-reduce set-up failed on ' + '; '.join(bad))

    def all_reduce_(self, t, mean=False, stream=None):
        """In place, on `stream` (default: the current stream): sum (mean: / world) over the ranks."""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        if t.numel() * 4 > self.max_bytes:
            raise ValueError(f'message of {t.numel() * 4} bytes > slot of {self.max_bytes}')
        s = stream if stream is not None else torch.cuda.current_stream(t.device)
        L.check(self.lib.yunet_allreduce(C.byref(self.comm), t.data_ptr(), t.numel(), 1 if mean else 0,
                                         C.c_void_p(s.cuda_stream)), 'yunet_allreduce')
        return t

    def status(self):
        """0, or the sequence number of the first call whose wait for a peer timed out (host read, no sync)."""
        return int(self.lib.yunet_comm_status(C.byref(self.comm)))

    def verify(self, n=12345, seed=7):
        """One message checked against the process group: every rank's input is all-gathered and summed in rank
        order on this rank -- the same order as the kernel's, so the comparison is exact."""
        g = torch.Generator().manual_seed(seed + self.rank)
        n = min(n, self.max_bytes // 4)
        x = torch.randn(n, generator=g).to(self.device)
        parts = [torch.empty_like(x) for _ in range(self.world)]
        if dist.get_backend(self.group) == 'nccl':
            dist.all_gather(parts, x, group=self.group)
        else:
            host = [torch.empty(n) for _ in range(self.world)]
            dist.all_gather(host, x.cpu(), group=self.group)
            parts = [h.to(self.device) for h in host]
        want = parts[0].clone()
        for p in parts[1:]:
            want += p
        # a self-check must not sit out the production time-out (minutes) if a peer's stores never become visible
        prev = L.set_option('oneshot_timeout_ms', 5000)
        try:
            got = self.all_reduce_(x.clone())
            torch.cuda.synchronize(self.device)
        finally:
            L.set_option('oneshot_timeout_ms', prev)
        ok = self.status() == 0 and torch.equal(got, want)
        flag = torch.tensor([1 if ok else 0])
        if dist.get_backend(self.group) == 'nccl':
            flag = flag.to(self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item())

    def close(self):
        for m in self._mapped:
            self.lib.yunet_comm_close(m)
        self._mapped = []
        if self._inbox:
            self.lib.yunet_comm_free(self._inbox, self._status)
            self._inbox = self._status = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
