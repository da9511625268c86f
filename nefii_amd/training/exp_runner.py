"""Command line of the Step-2 training run (code/training/exp_runner.py:10-130): same flags where they concern the hot
path; flags of subsystems this build does not have (plots, tensorboard, camera training, GPU auto-pick) are accepted
and ignored so that the reference's run scripts (robot/run_s2.sh, Physg_scripts/run_physg.sh) keep working."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--conf', type=str, required=True)
    p.add_argument('--data_split_dir', type=str, default='')
    p.add_argument('--data_split_dir_test', type=str, default='')
    p.add_argument('--gamma', type=float, default=1.0)
    p.add_argument('--subsample', type=int, default=1)
    p.add_argument('--batch_size', type=int, default=1)
    p.add_argument('--secondary_batch_size', type=int, default=1024)
    p.add_argument('--nepoch', type=int, default=2000)
    p.add_argument('--max_niter', type=int, default=200001)
    p.add_argument('--expname', type=str, default='')
    p.add_argument('--exps_folder', type=str, default='exps')
    p.add_argument('--old_expdir', type=str, default='')
    p.add_argument('--is_continue', default=False, action='store_true')
    p.add_argument('--timestamp', default='latest', type=str)
    p.add_argument('--checkpoint', default='latest', type=str)
    p.add_argument('--freeze_geometry', default=False, action='store_true')
    p.add_argument('--freeze_idr', default=False, action='store_true')
    p.add_argument('--freeze_decompose_render', default=False, action='store_true')
    p.add_argument('--freeze_light', default=False, action='store_true')
    p.add_argument('--freeze_diffuse', default=False, action='store_true')
    p.add_argument('--wo_mask', default=False, action='store_true')
    p.add_argument('--roughness_warmup', type=int, default=-1)
    p.add_argument('--specular_warmup', type=int, default=-1)
    p.add_argument('--secondary_train_interval', type=int, default=0)
    p.add_argument('--pretrain_geometry_path', type=str, default='')
    p.add_argument('--pretrain_idr_rendering_path', type=str, default='')
    p.add_argument('--pretrain_diffuse_path', type=str, default='')
    p.add_argument('--light_sg', type=str, default='')
    p.add_argument('--geometry', type=str, default='')
    p.add_argument('--geometry_neus', type=str, default='')
    p.add_argument('--local_rank', type=int, default=-1)
    p.add_argument('--model_class', type=str, default='nefii_amd.model.implicit_differentiable_renderer.IDRNetwork')
    p.add_argument('--dataset_class', type=str, default='',
                   help="default: the conf's train.dataset_class; nefii_amd.datasets.synthetic_dataset."
                        "SyntheticSceneDataset needs no data on disk")
    p.add_argument('--coordinate_type', type=str, default='mitsuba')     # exp_runner.py: axes of the logged envmap image
    p.add_argument('--no_graph', default=False, action='store_true')
    p.add_argument('--trace_tier', default=None, action='store_true',
                   help="tiered sphere tracing for every trace of this run (DESIGN.md 4f; default: the conf's train.trace_tier, "
                        "else off)")
    p.add_argument('--bracket_staged_eval', default=None, action='store_true',
                   help="stage the bracket search of the secondary traces behind the measured slope bound (DESIGN.md section 4; "
                        "bit-identical while the bound holds; default off)")
    p.add_argument('--plots', default=False, action='store_true',
                   help="every train.plot_freq iterations render one training view and write its buffers under plots/")
    opt, _ignored = p.parse_known_args(argv)
    from nefii_amd.training.idr_train import IDRTrainRunner
    local_rank = opt.local_rank if opt.local_rank > -1 else (int(os.environ['LOCAL_RANK']) if 'RANK' in os.environ else -1)
    runner = IDRTrainRunner(
        conf=opt.conf, data_split_dir=opt.data_split_dir, gamma=opt.gamma, subsample=opt.subsample,
        batch_size=opt.batch_size, secondary_batch_size=opt.secondary_batch_size, nepochs=opt.nepoch,
        max_niters=opt.max_niter, expname=opt.expname or 'default', exps_folder_name=opt.exps_folder,
        old_expdir=opt.old_expdir, is_continue=opt.is_continue, timestamp=opt.timestamp, checkpoint=opt.checkpoint,
        freeze_geometry=opt.freeze_geometry, freeze_idr=opt.freeze_idr,
        freeze_decompose_render=opt.freeze_decompose_render, freeze_light=opt.freeze_light,
        freeze_diffuse=opt.freeze_diffuse, wo_mask=opt.wo_mask, roughness_warmup=opt.roughness_warmup,
        specular_warmup=opt.specular_warmup, secondary_train_interval=opt.secondary_train_interval,
        pretrain_geometry_path=opt.pretrain_geometry_path, pretrain_idr_rendering_path=opt.pretrain_idr_rendering_path,
        pretrain_diffuse_path=opt.pretrain_diffuse_path, light_sg_path=opt.light_sg, geometry=opt.geometry, geometry_neus=opt.geometry_neus,
        local_rank=local_rank, model_class=opt.model_class, dataset_class=opt.dataset_class or None,
        graph=not opt.no_graph, plots=opt.plots, coordinate_type=opt.coordinate_type, trace_tier=opt.trace_tier, bracket_staged_eval=opt.bracket_staged_eval)
    runner.run()


if __name__ == '__main__':
    main()
