#!/bin/bash
# Conformance harness: runs the REFERENCE's own test files, in place under /root/reference, on top of this
# build's hot-path modules (tools/conformance/keisei_shim_plugin.py aliases keisei.training.* -> keisei_amd.training.*;
# everything else -- config, db, checkpoint, the unmodified katago_loop.py -- comes from the reference tree).
# Dev container only: nothing is copied into the repo, nothing travels to the GPU box.
#   tools/conformance.sh            # the hot-path files (CPU)
#   tools/conformance.sh -k gae     # extra pytest arguments are passed through
set -u
REF=${KEISEI_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
export PYTHONDONTWRITEBYTECODE=1
export KEISEI_CONFORMANCE=1
export PYTHONPATH="$HERE/conformance:$HERE/..${PYTHONPATH:+:$PYTHONPATH}"
FILES=(
  tests/test_gae.py tests/test_gae_batched.py tests/test_value_adapter.py tests/test_se_resnet.py
  tests/test_registries.py tests/test_registry_validation.py tests/test_entropy_annealing.py
  tests/test_model_degenerate_configs.py tests/test_models.py tests/test_model_variants.py
  tests/test_katago_ppo.py tests/test_split_merge_gae_opt.py tests/test_amp.py tests/test_torch_compile.py
  tests/test_pytorch_training_gaps.py tests/test_pytorch_amp_pipeline.py tests/test_katago_obs_channels.py
  tests/unit/test_distributed.py tests/unit/test_distributed_setup.py tests/unit/test_transformer_forward.py
  tests/test_katago_loop.py tests/test_katago_loop_integration.py tests/test_lr_scheduler.py
  tests/test_checkpoint.py tests/test_checkpoint_architecture.py tests/test_checkpoint_optimizer_state.py
  tests/test_sl_pipeline.py tests/test_sl_amp.py
  tests/integration/test_ddp_training.py
)
cd "$REF" || exit 2
if [ "${1:-}" = "--loop-helpers" ]; then
  # second leg: the reference's tests of its rollout helpers (and its loop) with keisei_amd.training.katago_loop's
  # split_merge_step / PendingTransitions / perspective corrections grafted into the reference's katago_loop module
  shift
  export KEISEI_CONFORMANCE_LOOP=1
  exec python -m pytest -p no:cacheprovider --noconftest -p keisei_shim_plugin -q tests/test_split_merge.py \
    tests/test_split_merge_transitions.py tests/test_split_merge_gae_opt.py tests/test_katago_loop.py "$@"
fi
exec python -m pytest -p no:cacheprovider --noconftest -p keisei_shim_plugin -q "${FILES[@]}" "$@"
