"""CPU restatement of the reference's word2vec-style recommenders (word2vec/skipgram_model.py,
cbow_model.py, linear_seq.py) -- TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (TF-1.0 graph, not
runnable here): built from the same restated pieces as oracle/ref_graph.py (lookups, scorer,
losses, TF-1.0 Adagrad), whose semantics are documented there.
"""
import numpy as np

from .ref_graph import Grads, RefEmbeddingAttribute


class RefW2VModel(object):
    def __init__(self, kind, size, batch_size, learning_rate, user_attributes, item_attributes,
                 item_ind2logit_ind, logit_ind2item_ind, n_input_items=1, loss_function='ce',
                 use_sep_item=True, output_feat=1, params=None, dtype=np.float64, top_N_items=100):
        assert kind in ('skipgram', 'cbow')
        user_attributes.set_model_size(size)                  # skipgram_model.py:35-40
        item_attributes.set_model_size(size)
        self.kind, self.batch_size, self.loss_function = kind, batch_size, loss_function
        self.learning_rate = float(learning_rate)
        self.n_input_items = n_input_items
        self.n_input = max(n_input_items, 1)                  # :74
        self.output_feat = output_feat
        self.top_N_items = top_N_items
        self.att_emb = RefEmbeddingAttribute(user_attributes, item_attributes, batch_size, None,
                                             self.n_input, use_sep_item, item_ind2logit_ind,
                                             logit_ind2item_ind, params=params, dtype=dtype)
        self.dt = self.att_emb.dt

    def prepare_warp(self, pos, pos_eval):
        self.att_emb.prepare_warp(pos, pos_eval)

    def _inputs(self, user_input, item_input):
        m = self.att_emb
        u, c_user = m.get_batch_user(user_input, concat=False)               # :80
        es, cs = [], []
        for i in range(self.n_input):
            feats, _, c = m.get_batch_item(item_input[i])                    # :83
            es.append(np.mean(np.stack(feats, 0), 0))                        # :84
            cs.append(c)
        return u, c_user, es, cs

    def step(self, user_input, item_input, item_output=None, forward_only=False, recommend=False):
        m, loss = self.att_emb, self.loss_function
        u, c_user, es, cs = self._inputs(user_input, item_input)
        n = self.n_input
        all_mean = np.mean(np.stack(es, 0), 0)
        if self.kind == 'skipgram':
            x_train, w_train = (u + es[0]) / 2, [1.0] + [0.0] * (n - 1)      # skipgram_model.py:87
        else:
            x_train, w_train = (u + all_mean) / 2, [1.0 / n] * n             # cbow_model.py:87-90
        x_test = u if self.n_input_items == 0 else (u + all_mean) / 2        # :91-99
        if recommend:
            logits, _ = m.get_prediction(x_test, 'full', self.output_feat)
            return np.argsort(-logits, axis=1, kind='stable')[:, :self.top_N_items].astype(np.int32)
        targets = m.target_mapping([item_output])[0]                         # linear_seq.py:76
        x = x_test if forward_only else x_train
        logits, c_pred = m.get_prediction(x, 'full', self.output_feat)
        mask = None if loss == 'ce' else m.mask(user_input, loss, None, forward_only=forward_only)
        bl, c_loss = m.compute_loss(logits, targets, loss, mask)
        if forward_only:
            return self.dt.type(bl.mean())                                   # loss_test (:126)
        mb = len(user_input)
        grads = Grads()
        d_logits, _ = m.compute_loss_bwd(c_loss, np.full((mb,), 1.0 / mb, dtype=self.dt))
        d_x = m.get_prediction_bwd(c_pred, d_logits, grads)
        for i in range(n):
            if w_train[i] == 0.0:
                continue
            nf = len(cs[i]['sites'])
            m.get_embedded_bwd(cs[i], [d_x * (0.5 * w_train[i] / nf)] * nf, None, grads)
        m.get_batch_user_bwd(c_user, d_x * 0.5, grads)
        m.apply_gradients(grads, self.learning_rate)
        return self.dt.type(bl.mean())
