"""Dev: end-to-end rate of the reference's training loop shape -- on-line synthesis of every batch on the
GPU (transform -> occluder -> flip -> HPR, train_cloudAAE_ycbv.py:96-117) followed by the train step."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloudaae_amd import tfrecord_io as TR, train_cloudAAE_ycbv as T
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
models, _ = TR.read_and_decode_obj_model(os.path.join(g, "obj_model_first1.tfrecords"))
obj = torch.from_numpy(np.repeat(models, 21, axis=0)).cuda()
recs = TR.PoseRecords([os.path.join(g, "pose_records_cls0_first4.tfrecords")])
for B, N in ((32, 512), (32, 256), (128, 512)):
    graph = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B}, replay=True)
    sel = np.arange(B) % 4
    x = {"translation": torch.from_numpy(recs.translation[sel]).cuda(), "axisangle": torch.from_numpy(recs.axisangle[sel]).cuda(),
         "class_id": torch.from_numpy(recs.class_id[sel]).cuda()}
    for i in range(3):
        graph.train_step(T.get_small_data(x, obj, seed=i))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20):
        graph.train_step(T.get_small_data(x, obj, seed=10 + i))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    el = T.get_small_data(x, obj, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20):
        graph.train_step(el)
    torch.cuda.synchronize(); ds = (time.perf_counter() - t0) / 20
    print("B=%d N=%d: synthesis + step %.2f ms per batch = %.0f clouds/s; step alone %.2f ms" % (B, N, dt * 1e3, B / dt, ds * 1e3), flush=True)
