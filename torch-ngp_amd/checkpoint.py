"""Checkpoint compatibility with the reference Trainer (nerf/utils.py:1015-1137).

File layout written by `Trainer.save_checkpoint`:  {'epoch', 'global_step', 'stats', 'model': state_dict,
['mean_count', 'mean_density' when cuda_ray], ['optimizer', 'lr_scheduler', 'scaler', 'ema' when full]}; a "best" checkpoint drops
`density_grid` from the model state; a bare state_dict is accepted as well.  The mirrored modules use the reference's parameter and
buffer names and shapes (encoder.embeddings / encoder.offsets, sigma_net.weights, color_net.weights, aabb_*, density_grid,
density_bitfield, step_counter), so the model state loads key for key; this module adds the Trainer-side bookkeeping and the
conversion between `torch.optim.Adam` + `GradScaler` state and `optim.NGPAdam`."""
import torch


def save_checkpoint(path, model, epoch=0, global_step=0, stats=None, optimizer=None, scaler=None, lr_scheduler=None, ema=None, full=False,
                    best=False, write=None):
    """`write`: whether THIS rank writes the file.  Default: rank 0 of an initialised process group (what the reference Trainer does,
    nerf/utils.py:650-655), every caller otherwise.

    With optim.NGPAdam(shard=True) the call is COLLECTIVE: every rank must make it (the fp32 master weights and the moments are gathered
    from their owners), and only the writing rank touches `path`.  A caller that keeps the reference's `if local_rank == 0:` guard
    around the call would leave rank 0 alone in a collective: that is refused with an error instead of a hang (pass write=... or call
    from every rank)."""
    import torch.distributed as dist
    in_group = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if write is None:
        write = (not in_group) or dist.get_rank() == 0
    state = {'epoch': epoch, 'global_step': global_step, 'stats': stats if stats is not None else {}}
    _materialize(model)
    if optimizer is not None and getattr(optimizer, 'shard', False):
        # optim.NGPAdam(shard=True): a rank keeps only its own 1/world of the fp32 master weights current between steps -- complete them
        # from their owners before ANY state_dict (also the model-only "best" checkpoints); a collective: every rank calls save_checkpoint
        if in_group:
            _all_ranks_here(optimizer)
        optimizer.wait_shadows()
        optimizer.gather_master()
    if getattr(model, 'cuda_ray', False):
        state['mean_count'] = model.mean_count
        state['mean_density'] = model.mean_density
    if full:
        if optimizer is not None:
            state['optimizer'] = optimizer.state_dict()
        if lr_scheduler is not None:
            state['lr_scheduler'] = lr_scheduler.state_dict()
        if scaler is not None:
            state['scaler'] = scaler.state_dict()
        if ema is not None:
            state['ema'] = ema.state_dict()
    sd = model.state_dict()
    if best and 'density_grid' in sd:  # nerf/utils.py:1066-1068
        sd = {k: v for k, v in sd.items() if k != 'density_grid'}
    state['model'] = sd
    if write:
        torch.save(state, path)
    return state


def _all_ranks_here(optimizer, timeout_s=60.0, use_store=None):
    """sharded checkpoints gather from every rank: a call made by a subset of the ranks (the reference's rank-0 guard) would hang in the
    first collective.  A roll call turns that into an error within `timeout_s`: `monitored_barrier` on gloo groups (the only backend that
    has it -- decided from the group's backend, not from exception text), and on every other backend (RCCL) a counter in the rendezvous
    store (TCPStore / FileStore: host side, independent of the collective library).  use_store: force either path (tests)."""
    import datetime
    import time
    import torch.distributed as dist
    group = getattr(optimizer, 'group', None)
    if use_store is None:
        use_store = dist.get_backend(group) != 'gloo'
    complaint = ('save_checkpoint with optim.NGPAdam(shard=True) is collective: every rank must call it (only the `write` rank '
                 'touches the file) -- ')
    if not use_store:
        try:
            dist.monitored_barrier(group=group, timeout=datetime.timedelta(seconds=timeout_s))
        except RuntimeError as e:
            raise RuntimeError(complaint + str(e)) from e
        return
    store = dist.distributed_c10d._get_default_store()
    world = dist.get_world_size(group)
    ranks = dist.get_process_group_ranks(group) if group is not None else list(range(world))
    # The round is agreed on THROUGH THE STORE, not counted per rank (a rank whose earlier call timed out or raised would be one
    # call ahead of the others for ever, and two optimizers on one group shared their keys): `epoch` names the round that is open; the rank
    # that completes a round moves the epoch on and deletes the round's counter.  A rank that arrives reads the epoch first, so a caller
    # that comes back after a failed round joins the round the others are in.
    base = f'ngp_checkpoint_roll_call/{ranks[0]}-{ranks[-1]}x{world}'
    epoch = int(store.add(base + '/epoch', 0))
    key = f'{base}/{epoch}'
    here = int(store.add(key, 1))
    if here >= world:                      # the last one in: open the next round, drop this one's counter behind the others' last look
        store.add(base + '/epoch', 1)
    deadline = time.monotonic() + timeout_s
    while True:
        if int(store.add(base + '/epoch', 0)) > epoch:
            if here >= world:
                try:
                    store.delete_key(key)
                except Exception:  # noqa: BLE001 -- a store without delete: the key stays (a few bytes per checkpoint)
                    pass
            return
        if time.monotonic() > deadline:
            seen = int(store.add(key, 0))
            store.add(key, -1)             # take this rank out of the round again: a later, complete call must not count it twice
            raise RuntimeError(complaint + f'{seen} of {world} rank(s) arrived within {timeout_s:.0f} s')
        time.sleep(0.002)


def load_checkpoint(checkpoint, model, optimizer=None, scaler=None, lr_scheduler=None, ema=None, model_only=False, map_location=None):
    """`checkpoint`: path or already-loaded dict.  Returns {'missing_keys', 'unexpected_keys', 'epoch', 'global_step', 'stats'}."""
    ck = torch.load(checkpoint, map_location=map_location, weights_only=False) if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, 'read') \
        else checkpoint
    info = {'missing_keys': [], 'unexpected_keys': [], 'epoch': None, 'global_step': None, 'stats': None}
    _materialize(model)   # (before the load: a table whose current copy is buffer set B would otherwise be copied OVER the loaded weights)
    if 'model' not in ck:  # a bare state_dict (nerf/utils.py:1089-1092)
        model.load_state_dict(ck)
        _after_model_load(model, optimizer)
        return info
    res = model.load_state_dict(ck['model'], strict=False)
    info['missing_keys'], info['unexpected_keys'] = list(res.missing_keys), list(res.unexpected_keys)
    if ema is not None and 'ema' in ck:
        ema.load_state_dict(ck['ema'])
    if getattr(model, 'cuda_ray', False):
        if 'mean_count' in ck:
            model.mean_count = ck['mean_count']
        if 'mean_density' in ck:
            model.mean_density = ck['mean_density']
    _after_model_load(model, optimizer)
    if model_only:
        return info
    info['stats'], info['epoch'], info['global_step'] = ck.get('stats'), ck.get('epoch'), ck.get('global_step')
    if optimizer is not None and 'optimizer' in ck:
        _load_optimizer(optimizer, ck['optimizer'], ck.get('scaler'))
    if lr_scheduler is not None and 'lr_scheduler' in ck:
        lr_scheduler.load_state_dict(ck['lr_scheduler'])
    elif optimizer is not None and 'lr_scheduler' in ck and getattr(optimizer, '_lr_lambda', None) is not None:
        # optim.NGPAdam is not a torch Optimizer (LambdaLR cannot wrap it): its own step-based rule resumes at the saved step.  Without a
        # rule the decayed rate itself was already restored from the optimizer state (load_torch_adam_state: lr / initial_lr)
        optimizer.schedule_step(int(ck['lr_scheduler'].get('last_epoch', 0)))
    if scaler is not None and 'scaler' in ck:
        scaler.load_state_dict(ck['scaler'])
    return info


def _materialize(model):
    """a hash table whose Adam sweep rides in the grid backward (optim.NGPAdam.enable_table_fusion) lives in two buffer sets; the torch
    Parameter is one of them -- make it the current one before the parameters are read or written from outside"""
    for p in model.parameters():
        fn = getattr(p, '_ngp_materialize', None)
        if fn is not None:
            fn()


def _after_model_load(model, optimizer):
    # the fp16 shadow copies an NGPAdam keeps next to the parameters must follow a load
    # (every rank loads the same complete file: nothing to gather in sharded mode)
    if optimizer is not None and hasattr(optimizer, '_sync_shadows'):
        optimizer._sync_shadows(assume_complete=True)
    elif optimizer is not None and hasattr(optimizer, 'sync_shadows'):
        optimizer.sync_shadows()


def _load_optimizer(optimizer, sd, scaler_sd):
    if hasattr(optimizer, 'load_torch_adam_state') and 'param_groups' in sd:  # a torch.optim.Adam state into optim.NGPAdam
        optimizer.load_torch_adam_state(sd, scaler_sd)
    else:
        optimizer.load_state_dict(sd)
