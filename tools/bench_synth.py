"""Times the GPU on-line synthesis (dev tool): samples/s of get_small_data and of the HPR kernel alone."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloudaae_amd import tfrecord_io as TR, train_cloudAAE_ycbv as T
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
models, _ = TR.read_and_decode_obj_model(os.path.join(g, "obj_model_first1.tfrecords"))
obj = torch.from_numpy(np.repeat(models, 21, axis=0)).cuda()
recs = TR.PoseRecords([os.path.join(g, "pose_records_cls0_first4.tfrecords")])
for B in (32, 128, 256):
    sel = np.arange(B) % 4
    x = {"translation": torch.from_numpy(recs.translation[sel]).cuda(), "axisangle": torch.from_numpy(recs.axisangle[sel]).cuda(),
         "class_id": torch.from_numpy(recs.class_id[sel]).cuda()}
    for _ in range(2):
        el = T.get_small_data(x, obj, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(5):
        el = T.get_small_data(x, obj, seed=i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("B=%d: %.2f ms per batch, %.0f samples/s (2 hulls per sample: 2449 + 2049 points); visible %.0f / %.0f" %
          (B, dt * 1e3, B / dt, float(el["num_vis_point"].float().mean()), float(el["num_vis_point_org"].float().mean())))
