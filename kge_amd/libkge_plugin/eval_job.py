"""EntityRankingJob with the fused rank-count kernel (eval.type: hip_entity_ranking)."""
import torch

from kge.job.eval_entity_ranking import EntityRankingJob

from .. import engine


class HipEntityRankingJob(EntityRankingJob):
    """Overrides only `_get_ranks_and_num_ties` (eval_entity_ranking.py:571-596): one
    streaming kernel pass instead of clone/isnan/isclose/gt/and/2x sum.  The loop,
    label handling, histograms and metrics stay the reference's code, so identical counts
    give identical MRR / Hits@k."""

    def _get_ranks_and_num_ties(self, scores: torch.Tensor, true_scores: torch.Tensor):
        if not scores.is_cuda:
            return super()._get_ranks_and_num_ties(scores, true_scores)
        if scores.stride(-1) != 1:
            scores = scores.contiguous()
        return engine.rank_counts(scores, true_scores.view(-1), atol=self.tie_atol, rtol=self.tie_rtol)
