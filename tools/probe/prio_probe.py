"""Stream priorities on this device (development aid): the range HIP reports (1 least .. -1 greatest) and whether a least-priority
stream can be created and wrapped for torch.  Round 3 ran the step with the parameter-gradient (side) stream at the least
priority: 515.4 / 514.5 imgs/s against 515.3 / 514.3 at the default - the workgroups of the big kernels own whole CUs (LDS) and
are not preempted, so the queue priority changes nothing here."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
rt = ctypes.CDLL('libamdhip64.so')
a, b = ctypes.c_int(0), ctypes.c_int(0)
torch.cuda.init()
print('rc', rt.hipDeviceGetStreamPriorityRange(ctypes.byref(a), ctypes.byref(b)), 'least', a.value, 'greatest', b.value)
st = ctypes.c_void_p()
print('create least-priority stream rc', rt.hipStreamCreateWithPriority(ctypes.byref(st), 1, a.value))
s = torch.cuda.ExternalStream(st.value)
p = ctypes.c_int(99)
print('wrapped for torch:', type(s).__name__, 'priority rc', rt.hipStreamGetPriority(ctypes.c_void_p(s.cuda_stream), ctypes.byref(p)), p.value)
hi = torch.cuda.Stream(priority=-1)
print('torch high-priority stream: rc', rt.hipStreamGetPriority(ctypes.c_void_p(hi.cuda_stream), ctypes.byref(p)), p.value)
