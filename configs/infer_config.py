"""Command-line surface of `python inference.py` — the reference's flags (its configs/infer_config.py:7-59; pinned option by
option to its parser in tests/golden/cli_flags.npz), same names, defaults and quirks, so existing run scripts keep working.  Only the diffusion group drives code in this repo; the
DUSt3R / render groups are forwarded to the reference implementation of those stages."""
import argparse


def get_parser():
    p = argparse.ArgumentParser()
    # ---- general
    p.add_argument("--image_dir", type=str, default="./test/images/fruit.png", help="input image, or a folder for sparse views")
    p.add_argument("--out_dir", type=str, default="./output", help="where results are written")
    p.add_argument("--device", type=str, default="cuda:0", help="compute device")
    p.add_argument("--exp_name", type=str, default=None, help="sub-folder of out_dir (default: timestamp + image name)")
    # ---- renderer / trajectory (handled by the reference's point-cloud stage)
    p.add_argument("--mode", type=str, default="single_view_txt",
                   help="single_view_target | single_view_txt | single_view_eval | sparse_view_interp")
    p.add_argument("--traj_txt", type=str, help="trajectory file: three lines of phi / theta / r")
    p.add_argument("--elevation", type=float, default=5., help="camera elevation in degrees (>0: from top)")
    p.add_argument("--center_scale", type=float, default=1., help="scale of the orbit centre depth")
    p.add_argument("--d_theta", nargs="+", type=int, default=10., help="target theta")
    p.add_argument("--d_phi", nargs="+", type=int, default=30., help="target phi")
    p.add_argument("--d_r", nargs="+", type=float, default=-.2, help="target radius change")
    p.add_argument("--d_x", nargs="+", type=float, default=0., help="pan right (+) / left, single_view_target")
    p.add_argument("--d_y", nargs="+", type=float, default=0., help="pan up (+) / down, single_view_target")
    p.add_argument("--mask_image", type=bool, default=False)
    p.add_argument("--mask_pc", type=bool, default=True)
    p.add_argument("--reduce_pc", default=False)
    p.add_argument("--bg_trd", type=float, default=0.)
    p.add_argument("--dpt_trd", type=float, default=1.)
    # ---- diffusion (this repo)
    p.add_argument("--ckpt_path", type=str, default="./checkpoints/model.ckpt", help="checkpoint path")
    p.add_argument("--config", type=str, default="./configs/inference_pvd_1024.yaml", help="model graph (yaml)")
    p.add_argument("--ddim_steps", type=int, default=50, help="DDIM steps")
    p.add_argument("--ddim_eta", type=float, default=1.0, help="eta (0.0 = deterministic)")
    p.add_argument("--bs", type=int, default=1, help="batch size, should be one")
    p.add_argument("--height", type=int, default=576)
    p.add_argument("--width", type=int, default=1024)
    p.add_argument("--frame_stride", type=int, default=10, help="fs conditioning value")
    p.add_argument("--unconditional_guidance_scale", type=float, default=7.5, help="classifier-free guidance scale")
    p.add_argument("--seed", type=int, default=123)
    p.add_argument("--video_length", type=int, default=25, help="frames (16 for the 16-frame model)")
    p.add_argument("--negative_prompt", default=False, help="unused")
    p.add_argument("--text_input", default=True, help="unused")
    p.add_argument("--prompt", type=str, default="Rotating view of a scene")
    p.add_argument("--multiple_cond_cfg", action="store_true", default=False)
    p.add_argument("--cfg_img", type=float, default=None)
    p.add_argument("--timestep_spacing", type=str, default="uniform_trailing")
    p.add_argument("--guidance_rescale", type=float, default=0.7)
    p.add_argument("--perframe_ae", default=True)
    p.add_argument("--n_samples", type=int, default=1)
    # ---- DUSt3R (reference stage)
    p.add_argument("--model_path", type=str, default="./checkpoints/DUSt3R_ViTLarge_BaseDecoder_512_dpt.pth")
    p.add_argument("--batch_size", default=1)
    p.add_argument("--schedule", type=str, default="linear")
    p.add_argument("--niter", default=300)
    p.add_argument("--lr", default=0.01)
    p.add_argument("--min_conf_thr", default=3.0)
    # ---- MI355X additions
    p.add_argument("--reference_root", type=str, default=None,
                   help="path of a Drexubery/ViewCrafter checkout providing DUSt3R, the PyTorch3D render and the CLIP "
                        "conditioners (default: $VIEWCRAFTER_REFERENCE)")
    p.add_argument("--renderings", type=str, default=None,
                   help="skip DUSt3R/render: a .pt/.npy file with point-cloud renders [T, H, W, 3] in [0, 1]")
    return p
