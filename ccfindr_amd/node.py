"""Host memory shared by the processes of ONE node (one process per GPU): segments under /dev/shm, mapped by name.

Eight MI355X share one host.  What the host computes once -- the ingestion of X, the two tiled layouts of a rank sweep --
or collects once -- the factor matrices of the sweep's units -- lives in a segment every process of the node maps,
instead of being rebuilt by every process or pickled through a process-group gather.  The reference ships the whole
``bundle`` to every MPI slave and gathers the per-run lists through Rmpi (reference R/bayesian.R:252-263); here the
data plane between the processes of a node is the node's own memory.

A segment is a file in /dev/shm created exclusively by its owner; peers open it by name (the name travels through the
process group as a tiny host object); the owner unlinks it as soon as every peer holds its mapping, so nothing is left
behind when a process dies afterwards.
"""
from __future__ import annotations

import mmap
import os
import socket
import uuid

import numpy as np


def shm_dir():
    d = os.environ.get("VBNMF_SHM_DIR") or "/dev/shm"
    if not os.path.isdir(d):
        import tempfile
        d = tempfile.gettempdir()
    return d


def free_bytes():
    """Bytes still available in the shared directory (a tmpfs that is full answers writes with SIGBUS, so callers ask first)."""
    forced = os.environ.get("VBNMF_TEST_SHM_FREE")          # tests: pretend the file system has this many bytes left
    if forced:
        return int(forced)
    try:
        st = os.statvfs(shm_dir())
        return int(st.f_bavail) * int(st.f_frsize)
    except OSError:
        return 0


def node_key():
    """What two processes must agree on to share /dev/shm: host name + the kernel's boot id.  ``VBNMF_NODE_KEY`` overrides
    (tests use it to rehearse the several-nodes path on one machine)."""
    forced = os.environ.get("VBNMF_NODE_KEY")
    if forced:
        return forced
    try:
        boot = open("/proc/sys/kernel/random/boot_id").read().strip()
    except OSError:
        boot = ""
    return f"{socket.gethostname()}:{boot}:{shm_dir()}"


def usable_cores():
    """Host cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota where one is set (a
    container with 128 visible CPUs and a quota of 16 runs 128 threads SLOWER than 32: the rank-sweep builder that took
    `cores // builders` threads from the affinity mask alone cut its layouts in 0.39 s instead of 0.18, profiles/r05_c4_sharded.json)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            text = open(path).read()
            if parse is not None:
                q = parse(text)
            else:
                quota = float(text)
                q = None if quota <= 0 else quota / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q is not None:
                n = min(n, max(1, int(q)))
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    return max(1, n)


def fresh_name(tag):
    return f"vbnmf_{tag}_{os.getpid()}_{uuid.uuid4().hex[:12]}"


def wait_for(name, timeout_s, failed_name=None):
    """Block until the file ``name`` exists in the shared directory (its owner renames it into place when complete);
    raises after ``timeout_s`` or when the owner left a failure note under ``failed_name``."""
    import time
    path = os.path.join(shm_dir(), name)
    t0 = time.perf_counter()
    while not os.path.exists(path):
        if failed_name is not None and os.path.exists(os.path.join(shm_dir(), failed_name)):
            try:
                msg = open(os.path.join(shm_dir(), failed_name)).read()
            except OSError:
                msg = "?"
            raise RuntimeError(f"the process that cuts the layouts failed: {msg}")
        if time.perf_counter() - t0 > timeout_s:
            raise TimeoutError(f"shared file {name} did not appear within {timeout_s:.0f} s")
        time.sleep(0.0005)
    return path


class Segment:
    """One shared mapping.  ``Segment.create(name, nbytes)`` (owner) / ``Segment.open(name)`` (peer)."""

    def __init__(self, name, mapping, size, owner):
        self.name, self.map, self.size, self.owner = name, mapping, size, owner
        self._unlinked = False

    @classmethod
    def create(cls, name, nbytes):
        path = os.path.join(shm_dir(), name)
        fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
        try:
            os.ftruncate(fd, max(int(nbytes), 1))             # sparse: pages appear as they are written
            mapping = mmap.mmap(fd, max(int(nbytes), 1))
        except BaseException:
            os.close(fd)
            os.unlink(path)
            raise
        os.close(fd)
        return cls(name, mapping, max(int(nbytes), 1), True)

    @classmethod
    def open(cls, name):
        path = os.path.join(shm_dir(), name)
        fd = os.open(path, os.O_RDWR)
        try:
            size = os.fstat(fd).st_size
            mapping = mmap.mmap(fd, size)
        finally:
            os.close(fd)
        return cls(name, mapping, size, False)

    def array(self, offset, shape, dtype=np.float64, order="F"):
        """A numpy view into the segment (it keeps the mapping alive for as long as the array lives)."""
        count = int(np.prod(shape))
        a = np.frombuffer(self.map, dtype=dtype, count=count, offset=int(offset))
        return a.reshape(shape, order=order)

    def unlink(self):
        if self.owner and not self._unlinked:
            self._unlinked = True
            try:
                os.unlink(os.path.join(shm_dir(), self.name))
            except FileNotFoundError:
                pass

    def close(self):
        self.unlink()
        try:
            self.map.close()
        except (BufferError, ValueError):                     # views are still alive: the mapping goes with the last of them
            pass
