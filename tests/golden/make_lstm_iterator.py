"""Golden call/yield sequence of the REAL reference lstm/data_iterator.py::DataIterator.next_sequence
(imports under py3; next_random uses xrange and cannot run) against a recording stand-in model.
Build container only:  python tests/golden/make_lstm_iterator.py -> tests/golden/lstm_iterator.json"""
import json
import os
import sys

REF = os.environ.get("ARX_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "lstm"))
import data_iterator as ref_di      # noqa: E402


class Recorder(object):
    """get_batch stand-in: bucket b holds sizes[b] sequences; returns its arguments."""

    def __init__(self, sizes, batch):
        self.sizes, self.batch, self.calls = sizes, batch, []

    def get_batch(self, data_set, bucket_id, start_id=None):
        self.calls.append(['train', bucket_id, start_id])
        return ('u', bucket_id, start_id), 'i', 'o', 'w', start_id + self.batch >= self.sizes[bucket_id]

    def get_batch_recommend(self, data_set, bucket_id, start_id=None):
        self.calls.append(['rec', bucket_id, start_id])
        return ('r', bucket_id, start_id), 'i', 'o', 'w', start_id + self.batch >= self.sizes[bucket_id]


def main():
    out = []
    for sizes, batch, stop, rec, take in (([5, 0, 9], 4, True, False, 100), ([3, 8], 4, False, False, 9),
                                          ([10], 5, True, True, 100)):
        m = Recorder(sizes, batch)
        it = ref_di.DataIterator(m, None, len(sizes), batch, [1.0])
        ys = []
        for k, y in enumerate(it.next_sequence(stop=stop, recommend=rec)):
            ys.append([list(y[0]), y[4]])
            if k + 1 >= take:
                break
        out.append({"sizes": sizes, "batch": batch, "stop": stop, "recommend": rec, "take": take,
                    "calls": m.calls, "yields": ys})
    with open(os.path.join(HERE, "lstm_iterator.json"), "w") as f:
        json.dump(out, f)
    print("ok", [len(c["yields"]) for c in out])


if __name__ == "__main__":
    main()
