"""CPU: refine -> annotation hand-off against golden files produced by executing the reference's
own exp/tools/result2ann.py and CocoDataset._det2json (oracle/handoff_oracle.py)."""
import copy
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle.handoff_oracle import synthetic_case
from pointtinybenchmark_amd import handoff

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'handoff.json')))


@pytest.mark.parametrize('name', sorted(GOLD))
def test_det2json_and_result2ann_match_reference_tool(name):
    case = GOLD[name]
    dataset, results, img_ids = synthetic_case(**case['kw'])
    det_json = handoff.det2json(results, img_ids, [1, 2])
    assert json.loads(json.dumps(det_json)) == case['det_json']
    out = handoff.result2ann(copy.deepcopy(dataset), copy.deepcopy(det_json), case['wh'])
    assert json.loads(json.dumps(out)) == case['out']


def test_refine_output_to_json_round_trip():
    """simple_test output (tensors) -> result dicts: per-class split, ann ids and geo survive; un-refined annotations keep
    their coarse box."""
    case = GOLD['geo_wh-1']
    dataset, results, img_ids = synthetic_case(**case['kw'])
    refine_out = []
    for per_cls in results:
        dets = torch.from_numpy(np.concatenate(per_cls))
        labels = torch.cat([torch.full((len(a),), i, dtype=torch.long) for i, a in enumerate(per_cls)])
        perm = torch.randperm(len(dets), generator=torch.Generator().manual_seed(0))
        refine_out.append((dets[perm], labels[perm]))
    got = handoff.refine_to_json(refine_out, img_ids, [1, 2])
    key = lambda d: (d['image_id'], d['ann_id'])
    assert sorted(got, key=key) == sorted(case['det_json'], key=key)
    out = handoff.result2ann(copy.deepcopy(dataset), copy.deepcopy(got), -1)
    refined = {d['ann_id'] for d in got}
    for a0, a1 in zip(dataset['annotations'], out['annotations']):
        assert a0['id'] == a1['id'] and a1['true_bbox'] == a0['true_bbox']
        assert (a1['bbox'] == a0['bbox']) == (a0['id'] not in refined)


def test_bbox2result_empty():
    r = handoff.bbox2result(np.zeros((0, 6), np.float32), np.zeros((0,), np.int64), 3)
    assert len(r) == 3 and all(a.shape == (0, 5) for a in r)


def test_cli(tmp_path):
    case = GOLD['plain_wh16']
    dataset, _, _ = synthetic_case(**case['kw'])
    a, b, c = tmp_path / 'ori.json', tmp_path / 'det.json', tmp_path / 'out.json'
    a.write_text(json.dumps(dataset))
    b.write_text(json.dumps(case['det_json']))
    subprocess.check_call([sys.executable, '-m', 'pointtinybenchmark_amd.handoff', '--ori_ann', str(a), '--det_file',
                           str(b), '--save_ann', str(c), '--wh', '16'], cwd=ROOT)
    assert json.load(open(c)) == case['out']
