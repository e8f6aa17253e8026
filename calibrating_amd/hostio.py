"""Device -> host hand-over for the NumPy-facing calls (the reference's surface is ndarray in, ndarray out:
stereo_camera.py:492-533 returns a dict of fresh arrays).

A result is copied into a page-locked block of torch's caching host allocator -- a fresh block per result, owned
by the returned array like any other ndarray and handed back to the cache when the array is collected -- with
asynchronous copies and ONE synchronisation at the end, instead of a blocking pageable copy per array.  ``Sink``
additionally moves each result on a side stream as soon as its kernel has been queued, so the copies of the early
results (the rectified images) run underneath the later kernels (SGBM).  At 1080p the ~60 MB a ``get_depth`` call
returns cost more wall time through pageable copies than all of its kernels.

Inputs are NOT staged: a pageable ``.cuda()`` of a 1080p RGB image takes 0.12 ms on the GPU box (50 GB/s), while a
fresh page-locked block per call costs 1.6 ms and two in a row 9.8 ms whenever the first is still in flight
(tools/microtests/upload_probe.py, profiles/r05_upload_probe.txt) -- the asymmetry to the results is that those are
large, many, and produced at different times.

``PINNED = False`` falls back to plain ``.cpu()`` copies (for hosts where page-locked memory is rationed); whatever a
call returns beyond ``PINNED_MAX_BYTES`` takes plain copies by itself (``to_host`` and ``Sink`` alike).
Ownership: every returned ndarray owns its page-locked block for as long as it lives (about 60 MB per 1080p
``get_depth`` call, rounded up to powers of two by the allocator and returned to its cache, not to the OS, on collection);
callers that keep many results around should copy what they keep (``np.array(x)``) or set ``PINNED = False``.
"""
PINNED = True
PINNED_MAX_BYTES = 1 << 30  # results larger than this in one call (big get_depth_batch calls) take plain copies

_side_streams = {}


def _side_stream(device):
    import torch
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def _start(t):
    import torch
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    return h


def to_host(*tensors):
    """ndarrays of CUDA tensors: all copies queued on the current stream, one synchronisation.  ONE tensor in -> the
    bare ndarray out; callers that zip the result with a variable number of keys use ``to_host_list``."""
    out = to_host_list(*tensors)
    return out[0] if len(out) == 1 else out


def to_host_list(*tensors):
    """``to_host`` without the single-result unwrapping: always a list, one ndarray per tensor."""
    import torch
    if not tensors:
        return []
    if not PINNED or sum(t.numel() * t.element_size() for t in tensors) > PINNED_MAX_BYTES:
        out = [t.cpu().numpy() for t in tensors]
    else:
        with torch.cuda.device(tensors[0].device):
            staged = [_start(t) for t in tensors]
            torch.cuda.current_stream().synchronize()
        out = [h.numpy() for h in staged]
    return out


class Sink:
    """Collects the results of one NumPy-facing call: ``send(key, tensor)`` as soon as the producing kernel is
    queued, ``collect()`` at the end -> {key: ndarray} (one synchronisation)."""

    def __init__(self, device):
        import torch
        self.device = device
        self.main = torch.cuda.current_stream(device)
        self.side = _side_stream(device) if PINNED else None
        self.staged = {}
        self.late = {}      # results beyond the PINNED_MAX_BYTES budget of this call: plain copies at collect()
        self.pinned_bytes = 0

    def send(self, key, t):
        import torch
        if not PINNED:
            self.staged[key] = t
            return
        nbytes = t.numel() * t.element_size()
        if self.pinned_bytes + nbytes > PINNED_MAX_BYTES:
            self.late[key] = t
            return
        self.pinned_bytes += nbytes
        self.side.wait_event(self.main.record_event())
        with torch.cuda.device(self.device), torch.cuda.stream(self.side):
            self.staged[key] = _start(t)
            self.done = self.side.record_event()  # everything sent so far has arrived once this event has passed
        t.record_stream(self.side)

    def collect(self, own_copies_only=False):
        """``own_copies_only``: wait for THIS call's copies alone (an event on the side stream), not for whatever was
        queued on the two streams afterwards -- what lets ``Stereo.get_depth_async`` keep several calls in flight."""
        if not PINNED:
            return {k: t.cpu().numpy() for k, t in self.staged.items()}
        if own_copies_only and getattr(self, "done", None) is not None:
            self.done.synchronize()
        else:
            self.side.synchronize()
            self.main.synchronize()
        out = {k: h.numpy() for k, h in self.staged.items()}
        out.update({k: t.cpu().numpy() for k, t in self.late.items()})
        return out


def bind_near_gpu(index=0):
    """Pin this process to the CPUs of the GPU's NUMA node (sysfs ``local_cpulist`` of its PCI function): page-locked
    blocks then come from the memory next to the GPU's PCIe root, and with one process per GPU every rank stays beside
    its own device.  Opt-in (a library does not change its host's affinity by itself); ``bench.py`` calls it.
    Returns the cpulist string, or None when sysfs does not tell."""
    import os
    try:
        import torch
        pr = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        cpulist = open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip()
        cpus = set()
        for part in cpulist.split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        if cpus:
            os.sched_setaffinity(0, cpus)
            return cpulist
    except Exception:
        pass
    return None
