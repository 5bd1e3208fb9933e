"""Test infrastructure (not product code): collects the pycocotools RLE strings the reference's own tests hold
(/root/reference/tests/data/test_coco_evaluation.py:24 -- three detections' "segmentation" dicts --
and /root/reference/tests/test_visualizer.py:54 -- one crowd annotation) into tests/golden/rle_coco.json.
They are outputs of pycocotools' maskApi.c (rleToString), i.e. known answers for the YTVIS result writer
(projects/SeqFormer/seqformer/data/ytvis_eval.py:196-202 encodes every result mask with mask_util.encode).

Run in the build container:  python oracle/make_golden_rle.py
"""
import ast
import json
import os

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rle_dicts(path):
    """every {"size": [...], "counts": "<str>"} literal in a python source file, in source order"""
    tree = ast.parse(open(path).read())
    found = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Dict):
            keys = [k.value for k in node.keys if isinstance(k, ast.Constant)]
            if set(keys) == {"size", "counts"}:
                d = ast.literal_eval(node)
                if isinstance(d["counts"], str):
                    found.append((node.lineno, {"size": list(d["size"]), "counts": d["counts"]}))
    return [d for _, d in sorted(found, key=lambda t: t[0])]


def main():
    out = []
    for rel in ("tests/data/test_coco_evaluation.py", "tests/test_visualizer.py"):
        for d in rle_dicts(os.path.join(REF, rel)):
            out.append({"source": rel, **d})
    assert len(out) == 4, len(out)
    dst = os.path.join(ROOT, "tests", "golden", "rle_coco.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(f"wrote {dst}: {len(out)} strings")


if __name__ == "__main__":
    main()
