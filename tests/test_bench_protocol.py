"""bench.py's multi-process protocol under two real processes with a stand-in engine (no GPU): RANK / WORLD_SIZE / MASTER_*
parsing, TCP rendezvous, weight-sync order (rank 0 packs, every rank joins the communicator, then the broadcast), barrier and
max-over-ranks timing, the per-step gather on every rank, ONE JSON line from rank 0 only.  The 8-GPU run of the driver is the
first place the real RCCL path meets more than one rank (RCCL refuses two ranks on one device), so the launcher logic is pinned
here."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, 'tests', 'helpers', 'run_bench_fake.py')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('world', [2, 3])
def test_bench_protocol_multi_rank(world):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, HELPER, '--gpus', str(world), '--steps', '3', '--warmup', '1', '--batch', '2',
                                       '--height', '16', '--width', '16'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    lines = [l for l in outs[0][0].splitlines() if l.strip()]
    assert len(lines) == 1, outs[0][0]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == world and rec['steps'] == 3 and rec['warmup'] == 1 and rec['scaling'] == 'weak'
    assert rec['config']['global_batch'] == 2 * world and rec['value'] > 0 and rec['unit'] == 'images/s'
    assert rec['cpu_baseline'] is None and rec['host_path'] is None          # rank 0, N = 1 only
    assert rec['roofline']['kernel'] == 'conv_wino' and 0 < rec['roofline']['frac'] <= 1
    for r in range(1, world):
        assert outs[r][0].strip() == '', "only rank 0 prints the JSON line"
    for r, (so, se) in enumerate(outs):
        log = [l for l in se.splitlines() if l.startswith('FAKELOG')][0]
        assert ('comm_init %d/%d' % (r, world)) in log and 'bcast' in log and 'comm_destroy' in log
        assert ('finalize' in log) == (r == 0)


@pytest.mark.parametrize('fail', ['all', '1'])
def test_bench_protocol_rccl_failure_degrades_to_tcp(fail):
    """If the RCCL communicator cannot be built (on every rank, or on one), ALL ranks switch together: own copy of the seeded
    weights, per-step gather over the rendezvous sockets; the JSON line still appears and says so."""
    world, port = 2, _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HP3D_FAKE_RCCL_FAIL=fail)
        procs.append(subprocess.Popen([sys.executable, HELPER, '--gpus', str(world), '--steps', '2', '--warmup', '1', '--batch', '2',
                                       '--height', '16', '--width', '16'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    rec = json.loads(outs[0][0].strip().splitlines()[-1])
    assert rec['n_gpus'] == world and rec['value'] > 0 and rec['config']['comm'].startswith('tcp-fallback')
    for r, (so, se) in enumerate(outs):
        log = [l for l in se.splitlines() if l.startswith('FAKELOG')][0]
        assert log.count('finalize') >= 1 and 'bcast' not in log.split('finalize')[-1]     # every rank ends on its own weights


@pytest.mark.parametrize('hang', ['all', '1', 'bcast'])
def test_bench_protocol_rccl_init_that_never_returns_has_a_deadline(hang):
    """ncclCommInitRank (or the first broadcast) blocking forever -- on every rank, on one, or after the communicator came up: the set-up
    calls run under HP3D_RCCL_TIMEOUT (hand3d_amd/dist.py), every rank agrees over the rendezvous, takes the TCP path and the JSON line
    appears well inside a minute saying so (the driver's first 8-rank run must not be able to hang)."""
    import time
    world, port = 2, _free_port()
    procs = []
    t0 = time.time()
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HP3D_FAKE_RCCL_HANG=hang, HP3D_RCCL_TIMEOUT='3')
        procs.append(subprocess.Popen([sys.executable, HELPER, '--gpus', str(world), '--steps', '2', '--warmup', '1', '--batch', '2',
                                       '--height', '16', '--width', '16'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert time.time() - t0 < 60
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    rec = json.loads(outs[0][0].strip().splitlines()[-1])
    assert rec['n_gpus'] == world and rec['value'] > 0
    assert rec['config']['comm'].startswith('tcp-fallback (TimeoutError: rccl init timeout'), rec['config']['comm']
    assert rec['config']['rccl_ranks'] == 0
    for r in range(1, world):
        assert outs[r][0].strip() == ''


def test_bench_protocol_single_process_plain():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, HELPER, '--steps', '2', '--warmup', '0', '--batch', '2', '--height', '16', '--width', '16',
                          '--cpu-seconds', '0'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec['n_gpus'] == 1 and rec['host_path'] is not None
    # the headline is the median of three K-step regions; SURVEY 8d's host-arrays-in -> keypoints-out rate from uint8 frames rides beside it
    assert rec['timed_regions'] == 3 and rec['value_min'] <= rec['value'] <= rec['value_max']
    assert rec['value_host_u8'] == rec['host_path']['value_host_u8'] > 0 and rec['host_path']['host_u8_batches_differ'] is True
    assert 'comm_init' not in out.stderr            # no launcher -> no communicator
    # BASELINE.json's other configurations ride on the same line (N = 1, full workload, float32), each with its own timed region
    oc = rec['other_configs']
    assert [c['config'] for c in oc] == ['C1', 'C2', 'C4-shard@240x320', 'C3-split', 'C5-shard'], oc
    for c in oc:
        assert 'error' not in c, c
        assert c['images_per_s'] > 0 and c['ms_per_step'] > 0 and c['dominant_family'] == 'conv_wino'
        assert c['parity_spot'] is None              # --cpu-seconds 0: no oracle leg, no spot check
    assert (oc[0]['batch'], oc[0]['height'], oc[0]['width']) == (1, 240, 320) and (oc[4]['batch'], oc[4]['height'], oc[4]['dtype']) == (128, 480, 'f16')
    assert oc[3]['options'] == {'wino4_split': 'auto'} and (oc[3]['batch'], oc[3]['height']) == (32, 320)
    assert rec['config']['cpu_affinity'] is None    # pinning is for launched ranks only


def test_bench_other_configs_can_be_switched_off():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, HELPER, '--steps', '1', '--warmup', '0', '--batch', '2', '--height', '16', '--width', '16',
                          '--cpu-seconds', '0', '--no-other-configs', '--no-host-path'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert 'other_configs' not in rec and rec['host_path'] is None


def test_bench_self_launch_spawns_the_ranks():
    """`python bench.py --gpus 3 ...` with NO launcher in the environment (the shape of the driver's N = 1 command): bench.py starts
    the three ranks itself, rank 0's JSON line comes out of the parent, and it says 3 GPUs and a 3-rank communicator."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'HP3D_RDZV_SECRET', 'HP3D_BENCH_ENTRY')}
    out = subprocess.run([sys.executable, HELPER, '--gpus', '3', '--steps', '3', '--warmup', '1', '--batch', '2', '--height', '16',
                          '--width', '16'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 3 and rec['config']['global_batch'] == 6 and rec['config']['comm'] == 'rccl'
    assert rec['config']['rccl_ranks'] == 3 and rec['value'] > 0
    logs = [l for l in out.stderr.splitlines() if l.startswith('FAKELOG rank')]
    assert sorted(l.split(':')[0] for l in logs) == ['FAKELOG rank 0', 'FAKELOG rank 1', 'FAKELOG rank 2'], out.stderr[-2000:]
    for l in logs:
        assert 'comm_init' in l and 'bcast' in l


def test_bench_self_launch_propagates_a_failing_rank():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'HP3D_RDZV_SECRET', 'HP3D_BENCH_ENTRY')}
    env['HP3D_FAKE_DIE_RANK'] = '1'
    env['HP3D_BENCH_GRACE_S'] = '3'          # (how long the survivors get before they are stopped: 20 s by default)
    out = subprocess.run([sys.executable, HELPER, '--gpus', '2', '--steps', '1', '--warmup', '0', '--batch', '2', '--height', '16',
                          '--width', '16'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert out.stdout.strip() == '', "no JSON line when a rank failed"


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, HELPER, '--gpus', '2', '--steps', '1', '--warmup', '0', '--batch', '2', '--height', '16',
                          '--width', '16'], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and 'WORLD_SIZE' in out.stderr and out.stdout.strip() == ''


def test_bench_launched_ranks_with_one_visible_device_each():
    """Per-rank HIP_VISIBLE_DEVICES isolation (or one rank per node): every rank sees ONE device and has LOCAL_RANK 0 while WORLD_SIZE
    is 2.  The device check of a launched rank is LOCAL_RANK < visible devices, not WORLD_SIZE <= visible devices."""
    world, port = 2, _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HP3D_FAKE_DEVICES='1')
        procs.append(subprocess.Popen([sys.executable, HELPER, '--gpus', str(world), '--steps', '2', '--warmup', '1', '--batch', '2',
                                       '--height', '16', '--width', '16'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    rec = json.loads(outs[0][0].strip().splitlines()[-1])
    assert rec['n_gpus'] == world and rec['config']['rccl_ranks'] == world and 'other_configs' not in rec


@pytest.mark.parametrize('launched', [False, True])
def test_bench_refuses_more_gpus_than_devices(launched):
    """`--gpus 8` on a box that shows 2 devices: one line on stderr and exit code 3 -- before any rank is started (plain call) or
    before the rank joins a rendezvous (under a launcher); nothing hangs, nothing is printed on stdout."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'HP3D_RDZV_SECRET', 'HP3D_BENCH_ENTRY')}
    env['HP3D_FAKE_DEVICES'] = '2'
    if launched:
        env.update(RANK='5', WORLD_SIZE='8', LOCAL_RANK='5', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    t0 = time.time()
    out = subprocess.run([sys.executable, HELPER, '--gpus', '8', '--steps', '1', '--warmup', '0', '--batch', '2', '--height', '16',
                          '--width', '16'], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 3 and out.stdout.strip() == ''
    assert 'only 2 HIP device(s)' in out.stderr and 'FAKELOG rank' not in out.stderr.replace('FAKELOG rank %s: \n' % env.get('RANK', '-'), '')
    assert time.time() - t0 < 60


def test_bench_family_accounting_of_the_profile_rows():
    """bench.py's roofline blocks are sums over per-launch profile rows (layer, kernel, ms, flops, bytes): the Winograd kernels with 4x4
    output tiles form ONE family whose EXECUTED flops are the algorithmic ones times the products each form really multiplies (F(4x4,3x3)
    36/144, a 7x7 filter as nine blocks 289/784, as four F(4x4,4x4) blocks 169/784); conv1_1's read pass (`conv_first_touch`, round 5)
    adds its TIME to the conv_first family but is neither a launch of that kernel nor algorithmic bytes -- so the family's HBM fraction
    cannot improve by hiding the pass."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows = [('a/conv1_1', 'conv_first_touch', 0.010, 0.0, 0.0), ('a/conv1_1', 'conv_first_3x3_c3', 0.170, 1e9, 8.78e8),
            ('b/conv1_1', 'conv_first_3x3_c3', 0.120, 1e9, 5.62e8),
            ('a/conv3_2', 'conv_wino4_f4x4_3x3', 0.5, 144e9, 1e8), ('b/conv6_2', 'conv_wino7_f4x4_4x4_as7x7', 0.1, 78.4e9, 1e7),
            ('b/conv6_3', 'conv_wino4_f4x4_3x3_as7x7_splitk', 0.2, 78.4e9, 1e7), ('x', 'conv_pw2_1x1_1x1', 0.05, 1e9, 1e6)]
    fam, total = bench.families(rows)
    assert abs(total - sum(r[2] for r in rows)) < 1e-12
    cf = fam['conv_first_3x3_c3']
    assert abs(cf[0] - 0.300) < 1e-12 and cf[3] == 2 and cf[2] == 8.78e8 + 5.62e8
    w4 = fam['conv_wino4']
    assert w4[3] == 3 and abs(w4[4] - (144e9 * 36 / 144 + 78.4e9 * 169 / 784 + 78.4e9 * 289 / 784)) < 1.0
    roof = bench.roof_of_family(fam, 'conv_first_3x3_c3', total, 157.3)
    assert roof['bound'] == 'hbm' and roof['launches'] == 2 and abs(roof['achieved'] - (8.78e8 + 5.62e8) / 0.300e-3 / 1e9) < 0.1
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
    r4 = bench.roof_of_family(fam, 'conv_wino4', total, 157.3)
    assert r4['bound'] == 'mfma' and abs(r4['achieved'] - w4[4] / 0.8e-3 / 1e12) < 0.01 and r4['achieved'] < r4['achieved_algorithmic']
