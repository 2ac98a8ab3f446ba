"""N>1 path on CPU: the batch shards by image index across ranks with no data-path collective; the only
cross-rank operations are a barrier and a MAX reduction of the elapsed time (bench.py).  world_size 2, gloo."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    import torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from cavif_rs_amd.synth import synth_image
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    B = 3
    mine = [rank * B + i for i in range(B)]                  # bench.py's shard rule
    sig = [int(synth_image(32, 24, index=i).astype('int64').sum()) for i in mine]
    got = [None] * world
    dist.all_gather_object(got, (mine, sig))
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        print(json.dumps({'shards': got, 'tmax': float(t[0])}))
    dist.destroy_process_group()
''') % ROOT


def test_two_rank_sharding(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29577')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                          '--master-port', '29577', str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    res = json.loads(line)
    idx = sorted(i for shard, _ in res['shards'] for i in shard)
    assert idx == list(range(6))                         # disjoint cover of the global batch
    sigs = [s for _, sg in res['shards'] for s in sg]
    assert len(set(sigs)) == 6                           # every rank encodes different images
    assert res['tmax'] == 2.0                            # MAX over ranks


GPU_WORKER = textwrap.dedent('''
    import os, sys, json, hashlib
    import torch.distributed as dist
    sys.path.insert(0, %r)
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    B = 2
    mine = [rank * B + i for i in range(B)]                  # bench.py's shard rule
    enc = m.Encoder().with_quality(80).with_speed(4).with_bit_depth(10).with_device(rank %% m.device_count())
    bt = m.BatchEncoder(enc, B, 192, 128, channels=3)
    for k, i in enumerate(mine):
        bt.upload(k, synth_image(192, 128, index=i))
    dist.barrier()
    bt.encode()
    out = [(i, hashlib.sha256(bt.get(k).avif_file).hexdigest()) for k, i in enumerate(mine)]
    got = [None] * world
    dist.all_gather_object(got, out)
    if rank == 0:
        print(json.dumps({'files': [x for g in got for x in g]}))
    bt.close()
    dist.barrier()
    dist.destroy_process_group()
''') % ROOT


import pytest


@pytest.mark.gpu
def test_two_ranks_encode_their_shards(oracle, tmp_path):
    """The N>1 path with the encoder in it: two ranks (sharing the box's GPU when it has one) encode disjoint image shards
    through the HIP path; every file == the oracle's for that global image index."""
    import hashlib, json
    from cavif_rs_amd.synth import synth_image
    script = tmp_path / 'gpu_worker.py'
    script.write_text(GPU_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29579', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                          '--master-port', '29579', str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert sorted(i for i, _ in res['files']) == [0, 1, 2, 3]
    for i, sha in res['files']:
        ref, _, _ = oracle.ravif_encode(synth_image(192, 128, index=i), quality=80, speed=4, depth=10)
        assert hashlib.sha256(ref).hexdigest() == sha


def test_bench_gpus_2_spawns_two_ranks_by_itself(oracle, tmp_path):
    """`python bench.py --gpus 2` with no launcher around it re-executes itself under torch.distributed.run (one rank per GPU; here both ranks share the
    SIMT emulator's one device): n_gpus == 2, every file of both ranks' slots == the oracle, both ranks' clocks in the line."""
    import hashlib, json
    from tests import emu
    from cavif_rs_amd.synth import synth_image
    w, h, B, slots = 64, 48, 1, 2
    man = {'config': {'width': w, 'height': h, 'speed': 4, 'quality': 80.0, 'bit_depth': 10},
           'sha256': [hashlib.sha256(oracle.ravif_encode(synth_image(w, h, index=i), quality=80, speed=4, depth=10)[0]).hexdigest() for i in range(2 * slots * B)]}
    mpath = tmp_path / 'manifest.json'
    mpath.write_text(json.dumps(man))
    env = dict(emu.env(), MI_BENCH_SHARE_DEVICES='1', MI_SYNTH_CACHE=str(tmp_path / 'cache'))
    env.pop('WORLD_SIZE', None); env.pop('RANK', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '0', '--batch', str(B), '--width', str(w), '--height', str(h),
                          '--pipeline', str(slots), '--no-cpu-baseline', '--no-pcie-loop', '--end-to-end', '0', '--manifest', str(mpath)], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert res['n_gpus'] == 2 and res['ms_per_step_per_rank']['ranks'] == 2
    assert res['ms_per_step_per_rank']['max'] <= res['ms_per_step'] * 1.001 + 1e-3
    assert res['output_identity']['checked'] == 2 * slots * B and res['output_identity']['equal'] == 2 * slots * B
    assert abs(res['value'] - 2 * B * w * h * 2 / 1e6 / (res['ms_per_step'] * 2 / 1e3)) < 1e-2 * res['value'] + 1e-3
