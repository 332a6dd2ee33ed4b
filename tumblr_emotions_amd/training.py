"""The training loop the reference delegates to `slim.learning.train`, shared by
train_image_model / train_text_model / train_deep_sentiment.

Behaviour kept from the reference (image_text_model/im_text_rnn_model.py:107-169):
  * `train_dir` is deleted and re-created (no resume)                                   :116-119
  * learning rate = initial_lr * decay_factor**epoch, re-assigned whenever
    step % (num_samples / batch_size) == 0                                             :137-147
  * a checkpoint every `save_interval_secs` (600 s) and at the end                      :160-167
  * log line `global step N: loss = L (S sec/step)` (slim train_step)                   A11
  * prints `Finished training. Last batch loss {:.3f}`                                  :169
The per-step work itself is SentimentNet.train_step (HIP kernels).
"""
import json
import os
import shutil
import time

import torch

from .synthetic import synthetic_batch_numpy, to_device


def _rank_world():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank(), torch.distributed.get_world_size()
    return 0, 1


def save_checkpoint(model, train_dir, step):
    """Variables under their TF names + Adam slots + step, as one torch file per checkpoint."""
    net = model.net
    torch.cuda.synchronize()
    path = os.path.join(train_dir, "model.ckpt-%d.pt" % step)
    # plain tensors / numbers / strings only, so that the file loads with weights_only=True
    torch.save({"variables": {k: torch.from_numpy(v) for k, v in net.state_dict().items()},
                "adam_m": net.store.m.cpu(), "adam_v": net.store.v.cpu(), "global_step": int(step),
                "config": json.dumps(model.config, default=str)}, path)
    with open(os.path.join(train_dir, "checkpoint"), "w") as f:
        json.dump({"model_checkpoint_path": os.path.basename(path), "global_step": step}, f)
    return path


def latest_checkpoint(checkpoint_dir):
    idx = os.path.join(checkpoint_dir, "checkpoint")
    if not os.path.exists(idx):
        return None
    with open(idx) as f:
        return os.path.join(checkpoint_dir, json.load(f)["model_checkpoint_path"])


def run_training(model, train_dir, num_steps, batch_fn=None, log_every=10, save_interval_secs=600,
                 dropout_mask=None, quiet=False):
    """model: ImageModel / TextModel / DeepSentiment front end (has .net, .config, .dataset)."""
    rank, world = _rank_world()
    if rank == 0:
        if os.path.exists(train_dir):
            shutil.rmtree(train_dir)           # "Delete old model", :116-118
        os.makedirs(train_dir)
    cfg = model.config
    batch_size, initial_lr, decay = cfg["batch_size"], cfg["initial_lr"], cfg["decay_factor"]
    # python-2 integer division, :140; under data parallelism one step consumes batch_size * world samples
    nb_batches = max(1, model.dataset.num_samples // (batch_size * world))
    net = model.net
    epoch, lr = 0, initial_lr
    last_save = time.time()
    loss_val = float("nan")
    t0 = time.time()
    for step in range(num_steps):
        if step % nb_batches == 0:                                   # :143-147
            lr = initial_lr * decay ** epoch
            if rank == 0 and not quiet:
                print("New learning rate: {0}".format(lr))
            epoch += 1
        model.learning_rate = lr
        batch = batch_fn(step) if batch_fn is not None else model.next_batch(step)
        model.labels = batch["labels"]
        net.train_step(batch, lr, dropout_mask=dropout_mask)
        model.logits = net.logits
        if (step + 1) % log_every == 0 or step + 1 == num_steps:
            loss_val = net.total_loss_value()                        # syncs: only on logging steps
            dt = (time.time() - t0) / (log_every if (step + 1) % log_every == 0 else max(1, (step + 1) % log_every))
            t0 = time.time()
            if rank == 0 and not quiet:
                print("global step %d: loss = %.4f (%.3f sec/step)" % (step + 1, loss_val, dt))
        if rank == 0 and time.time() - last_save >= save_interval_secs:
            save_checkpoint(model, train_dir, step + 1)
            last_save = time.time()
    if rank == 0:
        save_checkpoint(model, train_dir, num_steps)
        print("Finished training. Last batch loss {0:.3f}".format(loss_val))
    return loss_val


def load_checkpoint(model, path):
    ck = torch.load(path, map_location="cpu", weights_only=True)       # never unpickles arbitrary objects
    model.net.load_state_dict({k: v.numpy() for k, v in ck["variables"].items()})
    model.net.store.m.copy_(ck["adam_m"])
    model.net.store.v.copy_(ck["adam_v"])
    model.net.step = int(ck["global_step"])
    return ck["global_step"]


def run_evaluation(model, checkpoint_dir, log_dir, mode, num_evals, batch_fn=None, quiet=False):
    """What evaluate_* do with slim.evaluation.evaluation_loop + streaming_accuracy
    (im_text_rnn_model.py:171-207): restore the newest checkpoint of `checkpoint_dir`, run `num_evals`
    batches, accumulate accuracy = mean(argmax(logits) == labels).  The reference's loop then waits for
    the next checkpoint forever and writes TensorBoard summaries; here one pass is made, the result is
    returned, printed and appended to <log_dir>/<mode>/accuracy.jsonl.
    As in the reference the graph is built with is_training = (mode == 'train') (:65)."""
    path = latest_checkpoint(checkpoint_dir)
    if path is None:
        raise FileNotFoundError("no checkpoint in %s" % checkpoint_dir)
    step = load_checkpoint(model, path)
    is_training = mode == "train"
    correct = total = 0
    for i in range(num_evals):
        batch = batch_fn(i) if batch_fn is not None else model.next_batch(10 ** 6 + i)
        logits = model.net.predict(batch, is_training=is_training)
        model.logits, model.labels = logits, batch["labels"]
        correct += int((logits.argmax(dim=1) == batch["labels"]).sum().item())     # streaming_accuracy
        total += int(batch["labels"].shape[0])
    acc = correct / max(total, 1)
    out_dir = os.path.join(log_dir, mode)
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "accuracy.jsonl"), "a") as f:
        f.write(json.dumps({"global_step": step, "accuracy": acc, "num_evals": num_evals, "mode": mode}) + "\n")
    if not quiet:
        print("global step %d: accuracy = %.4f (%d samples, mode %s)" % (step, acc, total, mode))
    return acc


class SyntheticInput:
    """Mixin: the input side of the reference's model constructors.  If `config['dataset_dir']` holds a
    converted dataset (photos/train_valid_split.txt + tfrecords/tumblr_<mode>_*.tfrecord, the layout of
    datasets/convert_to_dataset.py:117-198) batches are read from it; otherwise they are synthetic."""

    def _init_input(self, config, post_size, vocab_size, nb_emotions, with_images, device):
        from .synthetic import SyntheticDataset
        self._in = (post_size, vocab_size, nb_emotions, with_images, device)
        self.post_ids = self.days = self.labels = None
        self._records = None
        ddir = config.get("dataset_dir")
        split = os.path.join(ddir or "", "photos", "train_valid_split.txt")
        if config.get("synthetic", False):
            self.dataset = SyntheticDataset(config.get("num_samples", 50000), nb_emotions)
        elif ddir and os.path.exists(split):
            from .datasets.convert_to_dataset import get_split_with_text
            self.dataset = get_split_with_text(config.get("mode", "train"), ddir)
        else:          # the reference fails in get_split_with_text; never train silently on made-up data
            raise IOError("no converted dataset under config['dataset_dir'] = %r (expected %s).  Set "
                          "config['synthetic'] = True for synthetic batches" % (ddir, split))

    def next_batch(self, step):
        post_size, vocab, nb, with_images, device = self._in
        rank, world = _rank_world()
        if hasattr(self.dataset, "data_sources"):            # real TFRecords
            if self._records is None:
                from .image_model.im_model import load_batch_with_text
                self._records = load_batch_with_text(self.dataset, self.config["batch_size"], height=224, width=224,
                                                     device=device, rank=rank, world=world, max_token_id=vocab,
                                                     num_classes=getattr(self.dataset, "num_classes", nb))
            b = next(self._records)
            if not with_images:
                b.pop("images")
        else:
            gb = self.config["batch_size"] * world
            b = synthetic_batch_numpy(gb, post_size, vocab, nb, seed=step, with_images=with_images)
            b = to_device(b, device, rank, world)
        self.post_ids, self.days = b["post_ids"], b["days"]
        return b
