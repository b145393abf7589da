import ctypes, time
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
p = ctypes.c_void_p()
hip.hipSetDevice(0)
hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20)); hip.hipFree(p)
for gb in (1, 8, 32, 48):
    t0 = time.time(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(gb << 30)); t1 = time.time()
    hip.hipMemset(p, 0, ctypes.c_size_t(gb << 30)); hip.hipDeviceSynchronize(); t2 = time.time()
    hip.hipFree(p); t3 = time.time()
    print("GB %d rc %d malloc %.3f s first-touch memset %.3f s free %.3f s" % (gb, rc, t1 - t0, t2 - t1, t3 - t2), flush=True)
