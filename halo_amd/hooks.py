"""Lightning-side hook for the sharded acquisition round.

The reference enters the round on rank 0 only while the other DDP ranks stall
(core/train_learners.py:307-326):

    def on_train_batch_start(self, batch, batch_idx):
        if self.local_rank == 0 and batch_idx in self.active_iters and not self.debug:
            <save checkpoint>; RegionSelection(cfg, feature_extractor, classifier, active_loader, active_round)
            self.log('active_round', ...); self.active_round += 1
        return batch, batch_idx

`sharded_on_train_batch_start` is that method with the rank gate removed from the selection:
every rank scores its contiguous block of the pool (halo_amd.pool.region_selection_sharded), the
pick tables are all-gathered, and all ranks leave together.  The checkpoint is still written by
rank 0 only.  Bind it with `use_sharded_rounds(SourceFreeLearner)` (or assign the method on any
learner class with the same attributes); nothing else in the learner changes.  Optional attributes
of the learner: `acquisition_group` (process group), `acquisition_driver` (test stand-in) and
`acquisition_global_budget` (None = reference behaviour; an integer G switches the round to the
pool-wide budget of halo_amd.pool.region_selection_sharded).  Contract of a custom `acquisition_driver`:
`driver(cfg, feature_extractor, classifier, loader, round_number) -> [(picks (n, 3), count)]` in loader
order; with a global budget it is additionally passed the keyword `write_files=False` and must then write
NO file (the files follow from the kept picks) -- a driver without that keyword is refused.  No Lightning import
is needed here: the method only touches attributes the reference's learner already has.
"""
import os

from .pool import region_selection_sharded


def sharded_on_train_batch_start(self, batch, batch_idx):
    if batch_idx in self.active_iters and not self.debug:
        if self.local_rank == 0:
            name = "model_before_round_{}.ckpt".format(self.active_round)
            print("\nSaving checkpoint: {}".format(name))
            self.trainer.save_checkpoint(os.path.join(self.cfg.SAVE_DIR, name))
            print(f"\n>>>>>>>>>>>>>>>> Active Round {self.active_round} (sharded) >>>>>>>>>>>>>>>>")
        self.last_round_tables = region_selection_sharded(self.cfg, self.feature_extractor, self.classifier,
                                                          self.active_loader, self.active_round,
                                                          group=getattr(self, "acquisition_group", None),
                                                          driver=getattr(self, "acquisition_driver", None),
                                                          # None (default) = the reference's per-image budget; an integer G =
                                                          # spend G regions over the whole pool this round (opt-in)
                                                          global_budget=getattr(self, "acquisition_global_budget", None))
        if self.local_rank == 0:
            self.log("active_round", self.active_round, on_step=True, on_epoch=False)
        self.active_round += 1          # on every rank: all of them ran the round
    return batch, batch_idx


def use_sharded_rounds(learner_cls):
    """Replace the learner's rank-0-only acquisition hook with the sharded one.  Returns the class."""
    learner_cls._reference_on_train_batch_start = learner_cls.__dict__.get("on_train_batch_start")
    learner_cls.on_train_batch_start = sharded_on_train_batch_start
    return learner_cls
