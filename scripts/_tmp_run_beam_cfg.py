import bench, argparse, torch, json, sys
for lanes in (1, 2):
    args = argparse.Namespace(batch=256, seconds=10.0, dec_streams=lanes)
    print(json.dumps(bench.espnet_beam_config(torch.device("cuda:0"), args)))
