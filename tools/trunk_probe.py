"""What grouping the two trunks' launches could buy for the reference's step shape (one image + 32 boxes; r06): the crop
trunk alone by batch, the full-image trunk alone, both on two streams (what DeviceNet.forward_images does) and on one.
The full-image map has 6080 pixels = 42 crops' worth: a launch pair grouped into one would at best behave like the crop
trunk at 32 + 42 = 74 crops.  usage: python tools/trunk_probe.py"""
import sys, os, torch, time
sys.path.insert(0, os.getcwd())
from monopsr_amd.core import device_net as dn, weights as W
dev=torch.device("cuda",0)
w=W.synthetic_weights(seed=0, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
net=dn.DeviceNet(w, device=dev, full_trunk=True)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
for B in (32,42,74,128,256):
    x=torch.randn((B,48,48,3),device=dev)*50
    print("crop trunk B=%d: %.3f ms"%(B,t(lambda: net.trunk(x,"crop"))))
img=torch.randn((1,160,608,3),device=dev)*50
print("full trunk 1 image: %.3f ms"%t(lambda: net.trunk(img,"full")))
img2=torch.randn((2,160,608,3),device=dev)*50
print("full trunk 2 images: %.3f ms"%t(lambda: net.trunk(img2,"full")))
x=torch.randn((32,48,48,3),device=dev)*50
side=torch.cuda.Stream()
def both():
    main=torch.cuda.current_stream(); side.wait_stream(main)
    with torch.cuda.stream(side): a=net.trunk(img,"full")
    b=net.trunk(x,"crop"); main.wait_stream(side); return a,b
print("both trunks, two streams: %.3f ms"%t(both))
def seq():
    return net.trunk(img,"full"), net.trunk(x,"crop")
print("both trunks, one stream: %.3f ms"%t(seq))
