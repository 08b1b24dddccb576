import subprocess, json, sys, os
for v in sys.argv[1:]:
    env = dict(os.environ)
    if v != "cur": env["FOCAL_HIP_LIB"] = f"focal_amd/lab/libfocal_hip_{v}.so"
    subprocess.run([sys.executable, "-m", "pytest", "tests/test_swt_parity_gpu.py", "-q", "-k", "train_step_loss_and_gradients and bf16"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    d = json.load(open("gpurun_out/observed_parity.json"))
    print(v, d.get("swt.train.loss.rank.bf16.abs_err_over_max1"))
