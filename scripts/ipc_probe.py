"""Can two processes on this pool share device memory through HIP IPC handles (what a device-side halo needs: the sender writes
the neighbour's receive slots directly)?  Parent allocates, exports the handle; the child opens it, writes a pattern with a
memset, closes; the parent reads the pattern back.  Prints one line per step; never spins on the device."""
import ctypes as C, multiprocessing as mp, sys

def hip():
    return C.CDLL("/opt/rocm/lib/libamdhip64.so")

def child(conn):
    h = hip()
    handle = conn.recv()
    buf = (C.c_char * 64).from_buffer_copy(handle)
    p = C.c_void_p()
    rc = h.hipIpcOpenMemHandle(C.byref(p), buf, 1)   # hipIpcMemLazyEnablePeerAccess
    print("child: hipIpcOpenMemHandle rc =", rc, flush=True)
    if rc == 0:
        rc2 = h.hipMemset(p, 0x5A, 4096)
        rc3 = h.hipDeviceSynchronize()
        print("child: memset rc =", rc2, "sync rc =", rc3, flush=True)
        print("child: close rc =", h.hipIpcCloseMemHandle(p), flush=True)
    conn.send(rc)

if __name__ == "__main__":
    mp.set_start_method("spawn")
    h = hip()
    p = C.c_void_p()
    print("parent: hipMalloc rc =", h.hipMalloc(C.byref(p), 4096), flush=True)
    h.hipMemset(p, 0, 4096); h.hipDeviceSynchronize()
    handle = (C.c_char * 64)()
    rc = h.hipIpcGetMemHandle(handle, p)
    print("parent: hipIpcGetMemHandle rc =", rc, flush=True)
    if rc != 0:
        sys.exit(0)
    a, b = mp.Pipe()
    pr = mp.Process(target=child, args=(b,))
    pr.start()
    a.send(bytes(handle))
    crc = a.recv()
    pr.join(60)
    out = (C.c_ubyte * 16)()
    h.hipMemcpy(out, p, 16, 2)
    print("parent: child rc =", crc, "first bytes after the child's write:", list(out)[:4], flush=True)
    print("IPC_OK" if crc == 0 and out[0] == 0x5A else "IPC_UNAVAILABLE", flush=True)
