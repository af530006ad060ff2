"""CPU: the launch scripts and launcher environments of the data-parallel path -- tools/dist_train.sh /
tools/slurm_train.sh (the reference's tools/dist_train.sh:11-21, tools/slurm_train.sh) and the launcher -> rank
environment mapping of init_dist (mmcv's init_dist for 'pytorch' | 'slurm' | 'mpi')."""
import os
import stat
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_python(tmp_path):
    """A `python` that prints its arguments, first on PATH: the scripts run up to the launch."""
    d = tmp_path / 'bin'
    d.mkdir()
    for name in ('python', 'srun'):
        f = d / name
        f.write_text('#!/bin/bash\necho "$0-ARGS: $@"\necho "IPC=$HSA_ENABLE_IPC_MODE_LEGACY"\n')
        f.chmod(f.stat().st_mode | stat.S_IEXEC)
    return dict(os.environ, PATH=f'{d}:{os.environ["PATH"]}')


def test_dist_train_sh_builds_the_reference_command_line(tmp_path):
    env = _fake_python(tmp_path)
    out = subprocess.run([os.path.join(ROOT, 'tools', 'dist_train.sh'), 'configs/yunet_n.py', '8', '29511',
                          '--work-dir', 'w d'], env=env, capture_output=True, text=True, check=True).stdout
    for part in ('-m torch.distributed.run', '--nnodes=1', '--node-rank=0', '--master-addr=127.0.0.1',
                 '--nproc-per-node=8', '--master-port=29511', 'tools/train.py configs/yunet_n.py --seed 0',
                 '--launcher pytorch --work-dir w d', 'IPC=0'):
        assert part in out, (part, out)
    # no port given: the third word is a train.py argument, the port defaults like the reference's (29500)
    out = subprocess.run([os.path.join(ROOT, 'tools', 'dist_train.sh'), 'configs/yunet_s.py', '2', '--auto-resume'],
                         env=dict(env, NNODES='2', NODE_RANK='1', MASTER_ADDR='10.0.0.1'),
                         capture_output=True, text=True, check=True).stdout
    for part in ('--nnodes=2', '--node-rank=1', '--master-addr=10.0.0.1', '--nproc-per-node=2', '--master-port=29500',
                 '--launcher pytorch --auto-resume'):
        assert part in out, (part, out)
    # usage error
    assert subprocess.run([os.path.join(ROOT, 'tools', 'dist_train.sh')], env=env, capture_output=True).returncode == 2


def test_slurm_train_sh(tmp_path):
    env = _fake_python(tmp_path)
    r = subprocess.run([os.path.join(ROOT, 'tools', 'slurm_train.sh'), 'part', 'job', 'configs/yunet_n.py', 'wd',
                        '--seed', '3'], env=dict(env, GPUS='16'), capture_output=True, text=True, check=True)
    for part in ('--partition=part', '--job-name=job', '--gres=gpu:8', '--ntasks=16', '--ntasks-per-node=8', '--cpus-per-task=5',
                 '--kill-on-bad-exit=1', 'tools/train.py configs/yunet_n.py --work-dir=wd --launcher=slurm --seed 3', 'IPC=0'):
        assert part in r.stdout, (part, r.stdout)
    # fewer GPUs than a node holds: one node, that many tasks
    r = subprocess.run([os.path.join(ROOT, 'tools', 'slurm_train.sh'), 'p', 'j', 'c.py', 'w'], env=dict(env, GPUS='4'),
                       capture_output=True, text=True, check=True)
    assert '--ntasks=4' in r.stdout and '--ntasks-per-node=4' in r.stdout and '--gres=gpu:4' in r.stdout
    assert subprocess.run([os.path.join(ROOT, 'tools', 'slurm_train.sh'), 'p', 'j'], env=env, capture_output=True).returncode == 2


def test_launcher_env_mapping():
    from yunet_amd.parallel import launcher_env
    # pytorch: the launcher exports everything; only the rendezvous defaults are filled in
    assert launcher_env('pytorch', env={}) == {'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29500'}
    assert launcher_env('pytorch', env={'MASTER_ADDR': 'h', 'MASTER_PORT': '1'}) == {}
    # slurm: task ids -> ranks, first host of the node list is the master
    e = {'SLURM_PROCID': '11', 'SLURM_NTASKS': '16', 'SLURM_LOCALID': '3', 'SLURM_NODELIST': 'gpu[07-08]'}
    got = launcher_env('slurm', env=e, first_host=lambda n: {'gpu[07-08]': 'gpu07'}[n])
    assert got == {'RANK': '11', 'WORLD_SIZE': '16', 'LOCAL_RANK': '3', 'MASTER_ADDR': 'gpu07', 'MASTER_PORT': '29500'}
    assert launcher_env('slurm', env=dict(e, MASTER_ADDR='x', MASTER_PORT='5'), first_host=None) == \
        {'RANK': '11', 'WORLD_SIZE': '16', 'LOCAL_RANK': '3'}
    # mpi
    mpi = {'OMPI_COMM_WORLD_RANK': '1', 'OMPI_COMM_WORLD_SIZE': '2', 'OMPI_COMM_WORLD_LOCAL_RANK': '1'}
    got = launcher_env('mpi', env=dict(mpi, MASTER_ADDR='head'))
    assert got['RANK'] == '1' and got['WORLD_SIZE'] == '2' and got['LOCAL_RANK'] == '1' and 'MASTER_ADDR' not in got
    with pytest.raises(KeyError):          # as mmcv's _init_dist_mpi: no silent 127.0.0.1 for a multi-node job
        launcher_env('mpi', env=mpi)
    with pytest.raises(ValueError):
        launcher_env('ssh', env={})


def test_first_slurm_host_without_scontrol():
    from yunet_amd.parallel import first_slurm_host
    assert first_slurm_host('node[01-04]') == 'node01'
    assert first_slurm_host('node[07,09-12],gpu3') == 'node07'
    assert first_slurm_host('a,b') == 'a' and first_slurm_host('gpu3') == 'gpu3'
    assert first_slurm_host('rack[2-3]n') == 'rack2n'
    for bad in ('', '[1-2]', 'n[a-b]'):
        with pytest.raises(ValueError):
            first_slurm_host(bad)
