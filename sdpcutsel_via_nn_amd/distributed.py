"""Candidate sets sharded over several GPUs (SURVEY.md section 8 e).

One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI).  Candidates are
independent, so each rank scores a contiguous shard with its own C-ABI handle and there is
exactly one exchange step per ranking: an all-gather of the per-shard head
(sel_size x (score fp64, global id int64) = 16 B per entry, <= 640 KB at 8 ranks) followed by a
replicated merge keyed (score descending, id ascending) -- the same order the reference's
stable sort gives on one list (cut_select_qp.py:601, :625, :653).  Counters are all-reduced.

The combined strategy's early-exit scan (cut_select_qp.py:606-623) is resolved through the
merged head of the STRONG class (positive and violated), see :meth:`ShardedSelector.select`.
:meth:`ShardedSelector.select_round` is the whole round (selection + eigen-cut rows of the own
candidates) with one collective and one host synchronisation (csrc/shard.hip).

With the "gloo" backend (CPU rehearsal of the choreography, used by the tests) the gathered
buffers are staged through host memory; the ranking and the merge still run on the device.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _capi

_BIG_M = 1000.0
_PAD_ID = np.iinfo(np.int64).max


class DeviceOps(object):
    """The two device operations the selector needs, on a :class:`_capi.Scorer`."""

    def __init__(self, scorer, device):
        self.scorer, self.device = scorer, device
        self._rec = {}
        # run the library on torch's current stream: its kernels, torch's copies and the
        # collectives torch enqueues are then ordered without host synchronisation
        scorer.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def ensure_scored(self, strat):
        """score what strategy ``strat`` ranks by and the current point does not have yet"""
        need = _capi.EIG if strat == 1 else _capi.NN if strat == 2 else (_capi.EIG | _capi.NN)
        have = self.scorer.get_stat(_capi.STAT_SCORED)
        if (have & need) != need:
            self.scorer.score(need & ~have)

    def local_head(self, strat, sel_size, count, want_secondary=False):
        """-> (scores[count] fp64, ids[count] int64, secondary[count] or None, n_total, counters);
        device tensors padded with (-inf, PAD, -inf).  secondary = obj_improve of each entry."""
        scores = torch.full((count,), float("-inf"), dtype=torch.float64, device=self.device)
        ids = torch.full((count,), _PAD_ID, dtype=torch.int64, device=self.device)
        sec = torch.full((count,), float("-inf"), dtype=torch.float64, device=self.device) if want_secondary else None
        tmp_s = torch.empty(count, dtype=torch.float64, device=self.device)
        tmp_i = torch.empty(count, dtype=torch.int64, device=self.device)
        w, total, _, counters = self.scorer.rank_device(strat, sel_size, count, tmp_i.data_ptr(), tmp_s.data_ptr())
        if w:
            scores[:w], ids[:w] = tmp_s[:w], tmp_i[:w]
            if want_secondary:
                tmp_o = torch.empty(w, dtype=torch.float64, device=self.device)
                self.scorer.gather_scores_device(w, tmp_i.data_ptr(), None, tmp_o.data_ptr())
                sec[:w] = tmp_o
        return scores, ids, sec, total, counters

    # -- fused round (sdpcut_shard_head_device / sdpcut_shard_finish_enqueue / _wait) ----------------
    max_head = 16384

    def shard_head(self, strat, count, into=None):
        """packed head record of this shard, enqueued on the current stream (no host sync); strat
        _capi.PART_COMBALL: three fields per entry (new score, id, obj_improve), else two.  into: an int64 device
        tensor of the record's length to write it to (part of a buffer several lists share)"""
        fields = 3 if strat == _capi.PART_COMBALL else 2
        rec = into
        if rec is None:
            rec = self._rec.get((count, fields))           # one buffer per record shape, reused every round
            if rec is None:
                rec = self._rec[(count, fields)] = torch.empty(8 + fields * count, dtype=torch.int64, device=self.device)
        self.scorer.shard_head_device(strat, count, rec.data_ptr())
        return rec

    def shard_finish_enqueue(self, world, count, allrec, sel_size, fields=2, pitch_words=0, offset_words=0):
        """merge of the gathered records + rows of the own entries, enqueued (no host sync); pitch / offset: this list's
        records inside a gathered buffer that several lists share"""
        self.scorer.shard_finish_enqueue(world, count, allrec.data_ptr() + 8 * int(offset_words), sel_size, fields, pitch_words)

    def shard_finish_wait(self):
        """-> dict(headers, idx, score, lam, coef, rhs, ks, pos, n_own): the round's one host wait"""
        return self.scorer.shard_finish_wait(own=True)

    def shard_finish(self, world, count, allrec, sel_size):
        """both halves in one call, rows at their head positions (ks = 0 / lam = NaN for other shards' entries)"""
        return self.scorer.shard_finish_round(world, count, allrec.data_ptr(), sel_size)

    def rows_of(self, global_ids):
        """eigen-cut rows of the entries of ``global_ids`` (numpy) that live on this shard
        -> (mine mask, lam, coef [., row_len], rhs, ks)"""
        sc = self.scorer
        mine = (global_ids >= sc.base) & (global_ids < sc.base + sc.N)
        lam, coef, rhs, _, ks = sc.cut_rows(global_ids[mine] - sc.base)
        return mine, lam, coef[:, :sc.row_len], rhs, ks

    def merge(self, scores, ids, count_out, secondary=None):
        out_s = torch.empty(count_out, dtype=torch.float64, device=self.device)
        out_i = torch.empty(count_out, dtype=torch.int64, device=self.device)
        self.scorer.merge_topk_device(scores.numel(), scores.data_ptr(), ids.data_ptr(), count_out,
                                      out_s.data_ptr(), out_i.data_ptr(),
                                      secondary.data_ptr() if secondary is not None else None)
        return out_s, out_i


class ShardedSelector(object):
    """Global top-``sel_size`` selection over candidate shards, replicated on every rank."""

    def __init__(self, ops, n_local, group=None):
        self.ops, self.group = ops, group
        self._gathered = {}
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # rehearsal on a one-GPU box (SDPCUT_FORCE_COLLECTIVES=1 under torch.distributed.run with one
        # rank): run the collectives even though there is nobody to talk to, so that the RCCL calls,
        # dtypes and stream hand-offs of the N > 1 path are exercised for real
        self._solo = self.world == 1 and not (dist.is_initialized() and os.environ.get("SDPCUT_FORCE_COLLECTIVES") == "1")
        self.n_local = int(n_local)
        self.n_global = self._sum(self.n_local)
        # rounds served by: the one-collective route / the every-entry-visited route (two collectives) / the unfused last resort
        self.path_counts = dict(common=0, comball=0, unfused=0)

    # -- collectives ---------------------------------------------------------------------
    def _sum(self, *vals):
        if self._solo:
            return vals[0] if len(vals) == 1 else list(vals)
        t = torch.tensor(vals, dtype=torch.int64)
        if dist.get_backend(self.group) == "nccl":
            t = t.to(self.ops.device)
        dist.all_reduce(t, group=self.group)
        out = [int(v) for v in t.cpu()]
        return out[0] if len(vals) == 1 else out

    def _all_gather(self, t):
        if self._solo:
            return t
        if dist.get_backend(self.group) == "nccl":
            key = (t.numel(), t.dtype)
            out = self._gathered.get(key)    # reused: the consumers are stream-ordered behind the collective
            if out is None:
                out = self._gathered[key] = torch.empty(self.world * t.numel(), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out, t, group=self.group)
            return out
        host = t.cpu()
        parts = [torch.empty_like(host) for _ in range(self.world)]
        dist.all_gather(parts, host, group=self.group)
        return torch.cat(parts).to(t.device)

    def _global_head(self, strat, sel_size, count, want_secondary=False):
        """Merged head of one ranking: (scores, ids, global list length, summed counters).

        ONE collective per head: every rank contributes a packed int64 record
        [total, nb_violated, strong, violated, nb_positive, 0, 0, 0 | scores | ids | secondary]
        (fp64 fields bit-cast), so the counters travel with the data instead of in a separate
        all-reduce; these messages are latency-bound (<= 120 KB per rank), fewer is faster."""
        s, i, sec, total, cnt = self.ops.local_head(strat, sel_size, count, want_secondary)
        head = [total, cnt["nb_violated"], cnt["strong"], cnt["violated"], cnt["nb_positive"]]
        if self.world == 1:
            g = head
        else:
            nf = 3 if sec is not None else 2
            rec = torch.empty(8 + nf * count, dtype=torch.int64, device=s.device)
            rec[:8] = torch.tensor(head + [0, 0, 0], dtype=torch.int64).to(s.device)
            rec[8:8 + count] = s.view(torch.int64)
            rec[8 + count:8 + 2 * count] = i
            if sec is not None:
                rec[8 + 2 * count:] = sec.view(torch.int64)
            allrec = self._all_gather(rec).view(self.world, -1)
            g = [int(v) for v in allrec[:, :5].sum(dim=0).cpu()]
            s_all = allrec[:, 8:8 + count].contiguous().view(-1).view(torch.float64)
            i_all = allrec[:, 8 + count:8 + 2 * count].contiguous().view(-1)
            sec_all = allrec[:, 8 + 2 * count:].contiguous().view(-1).view(torch.float64) if sec is not None else None
            s, i = self.ops.merge(s_all, i_all, count, sec_all)
        valid = min(count, g[0])
        return s[:valid], i[:valid], g[0], dict(nb_violated=g[1], strong=g[2], violated=g[3], nb_positive=g[4])

    # -- selection -----------------------------------------------------------------------
    def select(self, strat, sel_size):
        """-> dict(ids int64 tensor, scores fp64 tensor, new_strat, n_total, counters): the
        first min(sel_size, length) entries of the reference's rank list over ALL shards.

        Combined strategy: the reference walks the obj_improve-sorted list until it has seen
        sel_size "strong" (positive and violated) entries (cut_select_qp.py:606-623).
          1. merged per-shard heads of the STRONG class give the global number of strong
             entries and, if there are at least sel_size, the answer itself: the head of the
             re-sorted list is exactly those entries, +BIG_M, in obj_improve order;
          2. otherwise the scan visits every entry on every shard, so each shard's own combined
             ranking with an unreachable quota is a sub-list of the global one; heads are merged
             with obj_improve as secondary key (the second sort at :625 is stable w.r.t. the
             first at :601)."""
        sel_size = min(int(sel_size), self.n_global)
        if strat in (1, 2):
            s, i, total, cnt = self._global_head(strat, sel_size, sel_size)
            return dict(ids=i, scores=s, new_strat=strat, n_total=total, counters=cnt)
        if strat != 4:
            raise ValueError("strategy must be 1, 2 or 4")
        if sel_size == 0:
            e = torch.empty(0, device=self.ops.device)
            return dict(ids=e.to(torch.int64), scores=e.to(torch.float64), new_strat=4, n_total=self.n_global,
                        counters=dict(strong=0, violated=0))
        s, i, n_strong, _ = self._global_head(_capi.PART_STRONG, 0, sel_size)
        if n_strong >= sel_size:
            strong = violated = sel_size
            ids, scores = i, s + _BIG_M
        else:
            quota = self.n_global + 1                     # never reached: every entry is visited
            scores, ids, _, cnt = self._global_head(4, quota, sel_size, want_secondary=True)
            strong, violated = cnt["strong"], cnt["violated"]
        new_strat = 1 if strong / sel_size < violated / self.n_global else 4     # cut_select_qp.py:630
        return dict(ids=ids, scores=scores, new_strat=new_strat, n_total=self.n_global,
                    counters=dict(strong=strong, violated=violated))

    def select_round(self, strat, sel_size, copy=True):
        """Selection AND eigen-cut rows of one round (cut_select_qp.py:165-182) over all shards
        -> dict(ids, scores, mine, lam, coef, rhs, ks, new_strat, n_total, counters), numpy.

        ``ids`` / ``scores`` are the replicated global head; ``mine`` marks the entries whose
        candidate lives on this rank, and lam/coef/rhs/ks hold the rows of exactly those (in head
        order) -- every rank generates the rows of its own candidates, nothing else moves.  The
        row arrays are views of the handle's pinned host block (written by the device, compacted
        by the library): valid until the next round on this selector.

        Common regime (strategies 1 and 2; strategy 4 with at least sel_size strong candidates
        overall): ONE collective and ONE host synchronisation per round -- the shard's head and
        its counters are packed on the device, all-gathered, merged and turned into rows by
        library calls (shard_head / shard_finish_enqueue / shard_finish_wait).  Strategy 4 with fewer
        strong candidates: every entry is visited by the scan, and a second pass of the same three
        calls gathers the shards' combined rankings with obj_improve as the merge's secondary key
        (one more collective, one more wait; no torch glue).

        copy=False: ``ids`` / ``scores`` are views of that block as well (no host copies at all on the round's
        critical path; the +BIG_M of the combined strategy is added in place).

        The shard need not be scored beforehand: shard_head scores what the current point lacks (and,
        when nothing has been scored yet, lets the score kernels prepare the selection's first radix
        digit); :meth:`select` on its own expects the scores (``ops.ensure_scored``)."""
        return self.end_round(self.begin_round(strat, sel_size), copy=copy)

    def begin_round(self, strat, sel_size):
        """First part of :meth:`select_round`, up to and including the enqueued merge + rows: no host wait.
        -> token for :meth:`end_round`.  Two selectors (the two covers of a QCQP round) may both begin before either ends."""
        token = self.begin_head(strat, sel_size)
        if token[3]:
            self.finish_enqueue(token, self._all_gather(token[4]))
        return token[:4]

    def begin_head(self, strat, sel_size, into=None):
        """only the shard's packed head (into = a slice of a buffer shared with other lists) -> token incl. the record;
        the caller all-gathers and calls :meth:`finish_enqueue`"""
        if strat not in (1, 2, 4):
            raise ValueError("strategy must be 1, 2 or 4")
        sel = min(int(sel_size), self.n_global)
        ops = self.ops
        fused = 1 <= sel <= getattr(ops, "max_head", 0) and hasattr(ops, "shard_finish_enqueue")
        rec = ops.shard_head(_capi.PART_STRONG if strat == 4 else strat, sel, into=into) if fused else None
        return (strat, sel_size, sel, fused, rec)

    def finish_enqueue(self, token, allrec, pitch_words=0, offset_words=0):
        self.ops.shard_finish_enqueue(self.world, token[2], allrec, token[2], pitch_words=pitch_words, offset_words=offset_words)

    @staticmethod
    def record_words(sel):
        return 8 + 2 * int(sel)

    def _unpack(self, out, strat, sel, length, cnt, copy, big_m):
        valid = min(sel, length)
        if copy:
            ids, scores = out["idx"][:valid].copy(), out["score"][:valid] + (_BIG_M if big_m else 0.0)
        else:
            ids, scores = out["idx"][:valid], out["score"][:valid]
            if big_m:
                scores += _BIG_M
        w = out["n_own"]
        pos = out["pos"][:w]
        w = int(np.searchsorted(pos, valid))          # (pads beyond the list never carry rows; be explicit)
        mine = np.zeros(valid, dtype=bool)
        mine[pos[:w]] = True
        return dict(ids=ids, scores=scores, mine=mine, lam=out["lam"][:w], coef=out["coef"][:w], rhs=out["rhs"][:w],
                    ks=out["ks"][:w], counters=cnt)

    def end_round(self, token, copy=True):
        strat, sel_size, sel, fused = token
        ops = self.ops
        if fused:
            out = ops.shard_finish_wait()
            hdr = out["headers"]
            g = (hdr[0] if hdr.shape[0] == 1 else hdr.sum(axis=0)).tolist()
            length = int(g[0])
            # g[4] != 0: some shard's selection gave up (csrc/topk.hip), its record is void
            if int(g[4]) == 0 and (strat != 4 or length >= sel):
                cnt = dict(nb_violated=int(g[1]), nb_positive=int(g[2]))
                if strat == 4:
                    cnt.update(strong=sel, violated=sel)
                res = self._unpack(out, strat, sel, length, cnt, copy, strat == 4)
                res.update(new_strat=strat, n_total=self.n_global if strat != 1 else length)
                self.path_counts["common"] += 1
                return res
            if int(g[4]) == 0 and strat == 4 and hasattr(ops, "shard_finish_enqueue"):
                # fewer than sel strong candidates overall (`length` of them): the scan visits every entry
                # (cut_select_qp.py:606-623), each shard's combined ranking is a sub-list of the global one
                rec = ops.shard_head(_capi.PART_COMBALL, sel)
                ops.shard_finish_enqueue(self.world, sel, self._all_gather(rec), sel, fields=3)
                out = ops.shard_finish_wait()
                hdr = out["headers"]
                g2 = (hdr[0] if hdr.shape[0] == 1 else hdr.sum(axis=0)).tolist()
                if int(g2[4]) == 0:
                    strong, violated = length, int(g2[1])            # every violated entry is seen by the scan
                    cnt = dict(nb_violated=int(g2[1]), nb_positive=int(g2[2]), strong=strong, violated=violated)
                    res = self._unpack(out, strat, sel, self.n_global, cnt, copy, False)
                    res.update(new_strat=1 if strong / sel < violated / self.n_global else 4, n_total=self.n_global)     # :630
                    self.path_counts["comball"] += 1
                    return res
        # last resort (a selection that declared itself void, heads beyond the fused path's 16384): the unfused route
        self.path_counts["unfused"] += 1
        if hasattr(ops, "ensure_scored"):
            ops.ensure_scored(strat)
        res = self.select(strat, sel_size)
        ids = res["ids"].cpu().numpy()
        mine, lam, coef, rhs, ks = ops.rows_of(ids)
        return dict(ids=ids, scores=res["scores"].cpu().numpy(), mine=mine, lam=lam, coef=coef, rhs=rhs, ks=ks,
                    new_strat=res["new_strat"], n_total=res["n_total"], counters=res["counters"])


class ShardedQCQPRound(object):
    """The QCQP composition of a round (cut_select_qcqp.py:63-103) over candidate shards.

    Both covers of the instance are sharded lists, one :class:`ShardedSelector` (one library handle
    per rank) each: the cover of the objective's sparsity pattern is ranked with ``strat``
    (``:64-73``), the constraints-only cover with feasibility (``:75-77``), and the round takes
    ``(A + B)[0:sel_size]`` (``:79``).  Every rank holds the replicated head and generates the rows
    of the candidates it owns; the cut counts of ``:85-97`` are summed over the ranks.
    BASELINE.json configs[4] (q_50_*, 5-variable sub-problems, the constraints-only cover of
    1.4e6 candidates spread over the GPUs)."""

    def __init__(self, sel_obj, sel_cons):
        self.sel_obj, self.sel_cons = sel_obj, sel_cons
        self._buf = {}

    def round(self, strat, sel_size, vars_values):
        """-> dict(new_strat, is_obj bool[w], ids int64[w] (global position in the cover the entry
        comes from), scores fp64[w], nb_sdp_cuts, nb_opt_cuts, nb_cuts_combined,
        rows_obj / rows_cons: dict(mine, lam, coef, rhs, ks) of this rank's own entries)"""
        if strat not in (1, 2, 4):
            raise ValueError("strategy must be 1, 2 or 4")
        so, sc = self.sel_obj, self.sel_cons
        new_strat, n_obj, nb_opt = strat, 0, 0
        a = b = ta = tb = None
        have_a = so is not None and so.n_global > 0
        # Both covers' halves are enqueued before the first host wait.  How many entries of B the round takes depends on
        # the length of A's list (:79); under strategies 2 / 4 that is the objective cover's size, known beforehand,
        # under strategy 1 the number of its violated candidates: B is then asked for sel_size and trimmed.
        rest_bound = sel_size - (min(so.n_global, sel_size) if (have_a and strat != 1) else 0)
        want_b = rest_bound > 0 and sc is not None and sc.n_global > 0
        if have_a:
            so.ops.scorer.set_point(vars_values)
        if want_b:
            sc.ops.scorer.set_point(vars_values)
        sel_a = min(sel_size, so.n_global) if have_a else 0
        sel_b = min(rest_bound, sc.n_global) if want_b else 0
        one_buffer = (have_a and want_b and 1 <= sel_a <= getattr(so.ops, "max_head", 0) and 1 <= sel_b <= getattr(sc.ops, "max_head", 0)
                      and hasattr(so.ops, "shard_finish_enqueue") and so.world == sc.world and so.group is sc.group
                      and getattr(so.ops, "device", None) == getattr(sc.ops, "device", None))
        if one_buffer:
            # ONE all-gather for the round: the two covers' records side by side in one buffer (these messages are latency-bound)
            la, lb = so.record_words(sel_a), sc.record_words(sel_b)
            buf = self._buf.get((la, lb))
            if buf is None:
                buf = self._buf[(la, lb)] = torch.empty(la + lb, dtype=torch.int64, device=so.ops.device)
            ta = so.begin_head(strat, sel_size, into=buf[:la])
            tb = sc.begin_head(1, rest_bound, into=buf[la:])
            allrec = so._all_gather(buf)
            so.finish_enqueue(ta, allrec, pitch_words=la + lb, offset_words=0)
            sc.finish_enqueue(tb, allrec, pitch_words=la + lb, offset_words=la)
            ta, tb = ta[:4], tb[:4]
        else:
            if have_a:
                ta = so.begin_round(strat, sel_size)
            if want_b:
                tb = sc.begin_round(1, rest_bound)
        if ta is not None:
            a = so.end_round(ta)
            n_obj = len(a["ids"])                        # min(len(comb_obj), sel_size), :79
            new_strat = a["new_strat"]
            if strat != 1:
                nb_opt = int(np.count_nonzero(a["scores"] > _BIG_M))       # :85-88 (entries beyond the head carry no +BIG_M)
        rest = sel_size - n_obj
        if tb is not None:
            b = sc.end_round(tb)
            if len(b["ids"]) > rest:                     # strategy 1: A's list turned out longer than nothing
                w = int(np.count_nonzero(b["mine"][:rest]))
                b = dict(b, ids=b["ids"][:rest], scores=b["scores"][:rest], mine=b["mine"][:rest], lam=b["lam"][:w],
                         coef=b["coef"][:w], rhs=b["rhs"][:w], ks=b["ks"][:w])
            if rest <= 0:
                b = None
        ids = np.concatenate([a["ids"] if a else np.empty(0, np.int64), b["ids"] if b else np.empty(0, np.int64)])
        scores = np.concatenate([a["scores"] if a else np.empty(0), b["scores"] if b else np.empty(0)])
        is_obj = np.zeros(ids.shape[0], dtype=bool)
        # :90-92 counts the entries whose first field is an int -- the optimality / combined rankings' entries;
        # under strategy 1 both parts are feasibility entries (index sets)
        is_obj[:n_obj] = strat != 1
        # cuts: an entry produces one unless its lambda_min >= -1e-15 (:737-739); own rows only, summed
        own = 0
        for part in (a, b):
            if part is not None and len(part["lam"]):
                own += int(np.count_nonzero(part["lam"] < -1e-15))
        nb_cuts = (so or sc)._sum(own)
        rows = lambda p: None if p is None else {k: p[k] for k in ("mine", "lam", "coef", "rhs", "ks")}
        return dict(new_strat=new_strat, is_obj=is_obj, ids=ids, scores=scores, nb_sdp_cuts=nb_cuts, nb_opt_cuts=nb_opt,
                    nb_cuts_combined=int(is_obj.sum()), rows_obj=rows(a), rows_cons=rows(b))
