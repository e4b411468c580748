import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gmat_amd
lib = gmat_amd.load()
w,h=3840,2160
src=[torch.randint(0,256,(h,w*3),dtype=torch.uint8,device="cuda") for _ in range(8)]
dst=[torch.empty((w,h*3),dtype=torch.uint8,device="cuda") for _ in range(8)]
s=C.c_void_p(); lib.gmat_stream_create(C.byref(s))
t=C.c_void_p(); lib.gmat_timer_create(C.byref(t))
for i in range(8): lib.gmat_rotate_flip_smooth(src[i].data_ptr(), w*3, dst[i].data_ptr(), h*3, w, h, 3, s)
lib.gmat_stream_sync(s)
lib.gmat_timer_begin(t,s)
for i in range(64): lib.gmat_rotate_flip_smooth(src[i%8].data_ptr(), w*3, dst[i%8].data_ptr(), h*3, w, h, 3, s)
lib.gmat_timer_end(t,s); ms=C.c_float(); lib.gmat_timer_elapsed_ms(t,C.byref(ms)); print("fused rotate+flip+smooth 4K rgb24: %.2f us"%(ms.value/64*1e3))
