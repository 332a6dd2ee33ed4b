"""tumblr_emotions_amd -- the Deep Sentiment training path of anthonyhu/tumblr-emotions,
rebuilt MI355X-first: hand-written gfx950 HIP kernels behind a C ABI (include/ds_kernels.h,
libds_kernels.so), orchestrated from Python with PyTorch-ROCm providing device memory, streams,
autograd plumbing and torch.distributed (RCCL).

Front ends keep the reference's module paths and call signatures:
    tumblr_emotions_amd.image_model.im_model.train_image_model(checkpoints_dir, train_dir, num_steps)
    tumblr_emotions_amd.text_model.text_embedding.train_text_model(train_dir, num_steps)
    tumblr_emotions_amd.image_text_model.im_text_rnn_model.train_deep_sentiment(checkpoints_dir, train_dir, num_steps)
"""
__version__ = "0.1.0"
