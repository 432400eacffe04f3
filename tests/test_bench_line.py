"""bench.py's final line: the driver keeps only a tail of stdout, so the line must stay compact whatever the legs
return (round 3's 19.8 KB line was cut and did not parse).  CPU only; the canned result is round 3's full line."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def canned():
    return json.load(open(os.path.join(ROOT, 'profiles', 'r3', 'bench_r3.json')))


def test_line_is_compact_and_complete():
    detail = canned()
    assert len(json.dumps(detail)) > 15000                       # the canned result is the one that overflowed
    text = bench.compact_line(detail)
    assert '\n' not in text and len(text) < 4096 and len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'parity', 'detail'):
        assert key in line, key
    assert line['value'] == detail['value'] and line['ms_per_step'] == detail['ms_per_step']
    assert set(line['roofline']) >= {'bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms',
                                     'algorithmic_bytes_per_launch', 'decode_path_frac'}
    assert abs(line['roofline']['frac'] - line['roofline']['achieved'] / line['roofline']['peak']) < 1e-4
    assert set(line['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind', 'sample', 'fresh_instance_value',
                                         'all_cores_value', 'all_cores'}
    assert set(line['parity']) == {'images', 'poses', 'max_abs_delta', 'discrete_mismatches'}
    assert 'workload' in line['config'] and 'model' not in line['config']
    for name, leg in line['configs'].items():
        # five numbers per leg, plus (round 5) where a leg has them: the synchronous product path, the reference's CPU time
        assert {'value', 'ms_per_step', 'decode_ms', 'frac', 'parity_ok'} <= set(leg) <= {
            'value', 'ms_per_step', 'decode_ms', 'frac', 'parity_ok', 'sync_value', 'vs_ref_cpu', 'ref_cpu_ms', 'cpu_1thread'}, (name, leg)
    assert line['configs']['config4']['parity_ok'] is True
    assert line['configs']['config3']['value'] == detail['configs']['config3']['value']


def test_line_survives_failed_and_oversized_legs():
    detail = canned()
    detail['configs']['config3'] = {'error': 'RuntimeError(' + 'x' * 5000 + ')', 'leg_seconds': 1.0}
    line = json.loads(bench.compact_line(detail))
    assert len(line['configs']['config3']['error']) <= 80
    detail['configs'].update({'leg%d' % i: dict(detail['configs']['config4']) for i in range(200)})
    text = bench.compact_line(detail)
    assert len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    assert line.get('configs_dropped') is True and line['value'] == detail['value'] and line['roofline'] and line['cpu_baseline']


def test_line_without_optional_parts():
    detail = {k: v for k, v in canned().items() if k not in ('configs', 'cpu_baseline', 'parity', 'bf16_backbone')}
    line = json.loads(bench.compact_line(detail))
    assert line['cpu_baseline'] is None and line['parity'] is None and 'configs' not in line


def test_gpus_without_launcher_respawns(monkeypatch):
    """`--gpus N` with N > 1 and no WORLD_SIZE must not run one rank: it re-executes under torch.distributed.run."""
    seen = {}

    def fake_execv(exe, argv):
        seen['argv'] = argv
        raise SystemExit(0)
    monkeypatch.setattr(os, 'execv', fake_execv)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3'])
    with pytest.raises(SystemExit):
        bench.spawn_ranks(4)
    argv = seen['argv']
    assert argv[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in argv
    assert argv[argv.index('--nproc-per-node') + 1] == '4' and argv[argv.index('--master-addr') + 1] == '127.0.0.1'
    assert argv[-4:] == ['--gpus', '4', '--steps', '3']


def test_watchdog_prints_the_headline_when_a_leg_hangs(tmp_path):
    """A leg that never returns (round 4 lost 25 GPU-minutes to one) must not take the headline with it: after the
    soft limit the watchdog prints the compact line from what exists and exits 0; with nothing to print it exits 4."""
    import subprocess
    code = '''
import json, sys, time
sys.path.insert(0, %r)
import bench
bench.write_detail = lambda d: []            # (do not touch the checkout from a test)
if sys.argv[1] == "with":
    d = json.load(open(%r))
    d.pop("configs", None)
    bench._PARTIAL["line"] = d
bench.start_watchdog(0.3)
time.sleep(30)
''' % (ROOT, os.path.join(ROOT, 'profiles', 'r3', 'bench_r3.json'))
    script = tmp_path / 'hang.py'
    script.write_text(code)
    ok = subprocess.run([sys.executable, str(script), 'with'], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0, ok.stderr[-2000:]
    line = json.loads(ok.stdout.strip().splitlines()[-1])
    assert line['value'] == canned()['value'] and 'watchdog' in line and line['roofline'] and line['cpu_baseline']
    assert 'watchdog after' in ok.stderr and 'hang.py' in ok.stderr               # the traceback says where it hung
    bad = subprocess.run([sys.executable, str(script), 'without'], capture_output=True, text=True, timeout=120)
    assert bad.returncode == 4 and not bad.stdout.strip()
