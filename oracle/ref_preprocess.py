"""CPU restatement of the reference's attribute preprocessing -- TEST INFRASTRUCTURE ONLY
(imported by tests/; the product path is arx.utils.preprocess / arx.attributes.comb_attribute).

PARITY UNPINNED for this file: utils/preprocess.py imports TensorFlow's gfile and uses Python-2
constructs (cPickle, xrange, bytes/str mixing), so it cannot be run here; what follows restates
its algorithm entity by entity, in plain loops, from reading it.  (utils/load_data.py DOES import
under Python 3 -- the loader is pinned against it, tests/golden/ml1m_small_load_raw_data.json.)

Tie order: the reference sorts `sorted(counts, key=counts.get, reverse=True)` over a Python-2
dict; here the dict is insertion ordered, so equally frequent tokens keep first-appearance order.
"""
UNK_ID, START_ID = 0, 1
START_VOCAB = ['_UNK', '_START']


def _bag(cell):
    # preprocess.py:79-82,206-210: a multi-hot cell is str(cell).split(',')
    if isinstance(cell, list):
        return [str(t) for t in cell]
    return (cell if isinstance(cell, str) else str(cell)).split(',')


def vocab_het(inds, features, types, logits_size_tr, max_vocab, threshold, prefix):
    """create_dictionary (preprocess.py:52-119) -> {column: token list}."""
    out = {}
    for i, t in enumerate(types):
        if t > 1:
            continue
        counts = {}
        for u in inds:                                   # once per training interaction (:69)
            cell = features[u][i]
            for tok in ([str(cell)] if t == 0 else _bag(cell)):
                counts[tok] = counts.get(tok, 0) + 1
        ranked = sorted(counts, key=counts.get, reverse=True)          # :93
        max_size = logits_size_tr + 2 if (prefix == 'item' and i == 0) else max_vocab   # :95-101
        kept = [w for w in ranked if counts[w] >= threshold]           # :112-113
        out[i] = (START_VOCAB + kept)[:max_size]                       # :114-116
    return out


def vocab_mix(inds, features, max_vocab, threshold):
    """create_dictionary_mix (preprocess.py:121-166) -> token list."""
    uid, rest = {}, {}
    for u in inds:
        for tok in _bag(features[u][0]):
            d = uid if tok.startswith('uid') else rest                 # :138-141
            d[tok] = d.get(tok, 0) + 1
    ranked = list(uid) + sorted(rest, key=rest.get, reverse=True)      # :145-146
    kept = [w for w in ranked if (uid.get(w, rest.get(w)) >= threshold)]
    return (START_VOCAB + kept)[:max_vocab]


def tokenize(features, types, vocabs):
    """tokenize_attribute_map (preprocess.py:168-238); vocabs: {column: token list}."""
    cat, mul = [], []
    for i, t in enumerate(types):
        if t > 1:
            continue
        row = {w: k for k, w in enumerate(vocabs[i])}
        if t == 0:
            cat.append([row.get(str(f[i]), UNK_ID) for f in features] + [START_ID])    # :194-199
        else:
            vals, starts, lens = [], [0], []
            for f in features:
                ids = [row.get(tok, UNK_ID) for tok in _bag(f[i])]
                ids = [v for v in ids if v != UNK_ID] or [UNK_ID]      # :211-214
                vals.extend(ids)
                lens.append(len(ids))
                starts.append(starts[-1] + len(ids))
            vals.append(START_ID)                                      # :223-226
            lens.append(1)
            starts.append(starts[-1] + 1)
            mul.append((vals, starts, lens, max(lens[:-1]) if len(lens) > 1 else 0))
    return cat, mul


def index_mapping_het(item2fea0, n_items):
    """HET.index_mapping (comb_attribute.py:162-176)."""
    order = [i for i in range(n_items) if item2fea0[i] != 0]
    return order


def index_mapping_mix(i_inds, logits_size_tr):
    """MIX.index_mapping (comb_attribute.py:83-98)."""
    cnt = {}
    for i in i_inds:
        cnt[i] = cnt.get(i, 0) + 1
    return sorted(cnt, key=cnt.get, reverse=True)[:logits_size_tr]


def mix_bags(values, names, types):
    """MIX.mix_attr (comb_attribute.py:105-133): one comma-joined bag per entity."""
    out = []
    for r in values:
        v = []
        for j, t in enumerate(types):
            if t == 0:
                v.append(names[j] + str(r[j]))
            elif t == 1:
                v.extend(names[j] + s for s in str(r[j]).split(','))
        out.append(','.join(v))
    return out


def full_mulhot(mul_entry, order):
    """filter_mulhot (preprocess.py:257-326) from a tokenised column: bags of the logit items."""
    vals, starts, lens, _ = mul_entry
    v_tr, seg_tr, len_tr = [], [], []
    for j, it in enumerate(order):
        v_tr.extend(vals[starts[it]:starts[it] + lens[it]])
        seg_tr.extend([j] * lens[it])
        len_tr.append([float(lens[it])])
    return v_tr, seg_tr, len_tr
