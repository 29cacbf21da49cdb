"""
Host placement for the streaming CLI: keep a rank's reader / writer threads and its pinned
staging slabs on the NUMA node its GPU hangs off (--numa auto).

The reference's parallel mode is one CPU process per scp shard with no notion of placement
(scripts/run_adapt_beamformer.sh:69-92); with one rank per GPU on a two-socket host half of
the ranks would otherwise copy every utterance across the socket interconnect twice (page
cache -> pinned slab by a thread on the far socket, then the DMA out of a slab the far
socket's memory controller owns).

bind(ctx, mode):  mode "auto": the node of ctx's device (sysfs numa_node of its PCI bus id);
an integer: that node; "off": nothing.  Sets the CPU affinity of the CALLING thread -- call it
before the thread pools and the pinned slabs are created: threads inherit the mask, and
hipHostMalloc'd / first-touched pages land on the node of the thread that touches them
(set_mempolicy(MPOL_PREFERRED) makes that explicit where the syscall is allowed).
Returns a dict for the run summary; never raises (a container without sysfs NUMA
information just stays unbound).
"""
import ctypes
import os


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def parse_cpulist(text):
    cpus = set()
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def node_count():
    online = _read("/sys/devices/system/node/online")
    return len(parse_cpulist(online)) if online else 1


def device_node(ctx):
    """(pci bus id, NUMA node or None) of the context's device."""
    try:
        bus = ctx.pci_bus_id()
    except Exception:  # noqa: BLE001
        return None, None
    node = _read(f"/sys/bus/pci/devices/{bus}/numa_node")
    try:
        node = int(node)
    except (TypeError, ValueError):
        node = None
    if node is not None and node < 0:
        node = None  # the kernel's "no affinity" answer (single-node hosts, some VMs)
    return bus, node


def _set_mempolicy_preferred(node):
    """set_mempolicy(MPOL_PREFERRED, {node}) for the calling thread (inherited by threads it
    starts); False if the syscall is not available (seccomp'd containers)."""
    import platform
    # the syscall number is per architecture (238 is migrate_pages on aarch64): known tables only
    SYS_set_mempolicy = {"x86_64": 238, "aarch64": 237}.get(platform.machine())
    if SYS_set_mempolicy is None:
        return False
    MPOL_PREFERRED = 1
    try:
        libc = ctypes.CDLL(None, use_errno=True)
        nbits = max(64, node + 1)
        mask = (ctypes.c_ulong * ((nbits + 63) // 64))()
        mask[node // 64] |= 1 << (node % 64)
        r = libc.syscall(SYS_set_mempolicy, MPOL_PREFERRED, ctypes.byref(mask), nbits + 1)
        return r == 0
    except Exception:  # noqa: BLE001
        return False


def bind(ctx, mode="auto"):
    info = {"mode": str(mode), "nodes": node_count(), "bound": False}
    if str(mode) == "off":
        return info
    if str(mode) == "auto":
        bus, node = device_node(ctx)
        info["pci_bus_id"] = bus
    else:
        try:
            node = int(mode)
        except ValueError:
            info["error"] = f"--numa {mode!r}: expected auto, off or a node number"
            return info
    info["node"] = node
    if node is None or info["nodes"] <= 1:
        return info  # nothing to choose between
    cpus = parse_cpulist(_read(f"/sys/devices/system/node/node{node}/cpulist"))
    try:
        allowed = os.sched_getaffinity(0)
    except OSError:
        allowed = set()
    cpus &= allowed
    if not cpus:
        info["error"] = f"node {node} has no CPU this process may run on"
        return info
    try:
        os.sched_setaffinity(0, cpus)
    except OSError as e:
        info["error"] = f"sched_setaffinity: {e}"
        return info
    info["bound"] = True
    info["cpus"] = len(cpus)
    info["mempolicy_preferred"] = _set_mempolicy_preferred(node)
    return info
