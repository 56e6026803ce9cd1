"""A/B of two BUILDS of the library on the training step (graph replay and eager), alternating child processes:
    python tools/step_lib_ab.py PAIRS[,PAIRS...] LIB_A LIB_B [LIB_C ...] [reps]
Each child loads one library through GRAFP_HIP_LIB, builds the model from the same seed, and times 20 replayed steps
(Trainer.step_graph) after capture; the parent alternates A, B, A, B ... so that box drift hits both alike.
(The experiment builds come from `make measure XFLAGS=-D... MLIB=../libgrafp_hip_x_NAME.so MDIR=_obj_x_NAME`.)"""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))


def child(pairs_list):
    import torch
    sys.path.insert(0, os.path.dirname(HERE))
    from grafp_amd.train import Trainer, build_model, synthetic_batch
    from grafp_amd.util import load_config
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    for pairs in pairs_list:
        cfg = load_config()
        cfg["bsz_train"] = pairs
        torch.manual_seed(1234)
        model = build_model(cfg, device=dev)
        tr = Trainer(cfg, model, dev, amp_dtype=torch.bfloat16)
        x_i, x_j = synthetic_batch(pairs, 7, dev)
        out = []
        for name, step in (("graph", tr.step_graph), ("eager", tr.step)):
            for _ in range(4):
                loss = step(x_i, x_j)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(20):
                    loss = step(x_i, x_j)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
            out.append(f"{name} {best:8.3f} ms")
        print(f"pairs={pairs:5d} {os.path.basename(os.environ.get('GRAFP_HIP_LIB', 'default'))}: " + "  ".join(out) +
              f"  loss {float(loss):.5f}", flush=True)
        del tr, model
        torch.cuda.empty_cache()


def main():
    if sys.argv[1] == "--child":
        return child([int(v) for v in sys.argv[2].split(",")])
    pairs, libs, reps = sys.argv[1], sys.argv[2:], 3
    if libs[-1].isdigit():
        reps, libs = int(libs[-1]), libs[:-1]
    for _ in range(reps):
        for lib in libs:
            env = dict(os.environ, GRAFP_HIP_LIB=os.path.abspath(lib))
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", pairs], env=env, check=True)


if __name__ == "__main__":
    main()
