"""MNIST generated-label accuracy: mnist/utils.py:273-306 around a classifier the caller supplies.

The reference imports a frozen MNIST classifier from ./mnist_dcnn/graph_optimized.pb (utils.py:276-290: input `x`, dropout
`Placeholder` fed 1.0, output `pred_class`); that file is not part of the checkout.  What the checkout does fix is the bookkeeping
around it, restated here:

  samples [draws, 100, 28, 28, 1]: `draws` sampler calls on the SAME 100 labels, ten per class in class order (model.py:277-282,
  473-484).  utils.py:292-295 regroups them to [10 classes, draws * 10, 28, 28, 1] -- class c = samples[:, 10c : 10c + 10] of every
  draw, draw-major -- walks each class in batches of 100 (an incomplete tail is dropped, :299), takes the batch accuracy against the
  class index, and returns the mean of the batch accuracies (:300-305).

`predict_fn(images)` takes float [100, 28, 28, 1] (the sampler's [0, 1] output range) and returns 100 class indices."""
import numpy as np

NUM_TEST = 100                                     # utils.py:288


def regroup_by_class(samples):
    """utils.py:292-295."""
    samples = np.asarray(samples)
    if samples.ndim != 5 or samples.shape[1] != 100:
        raise ValueError("samples: [draws, 100, H, W, C] (ten samples per class in class order), got %s" % (samples.shape,))
    return (samples.transpose((1, 0, 2, 3, 4)).reshape((10, 10) + samples.shape[:1] + samples.shape[2:])
            .reshape((10, -1) + samples.shape[2:]))


def generated_label_accuracy(dataset, samples, predict_fn):
    """utils.py:273-306.  Raises for any dataset but 'mnist', as the reference does, and when no classifier is given."""
    if dataset != 'mnist':
        raise ValueError('generated label acc only implemented for mnist')
    if predict_fn is None:
        raise RuntimeError("generated label accuracy: no classifier.  The reference reads ./mnist_dcnn/graph_optimized.pb "
                           "(mnist/utils.py:276), which the checkout does not hold; pass predict_fn "
                           "(train_mnist.py --label_classifier_fn module:callable)")
    test_images = regroup_by_class(samples)
    acc_sum, num_sum = 0.0, 0
    for y_actual, class_samples in enumerate(test_images):
        for ii in range(NUM_TEST, class_samples.shape[0] + 1, NUM_TEST):
            y = np.asarray(predict_fn(class_samples[ii - NUM_TEST:ii])).reshape(-1)
            if y.shape[0] != NUM_TEST:
                raise ValueError("predict_fn returned %d predictions for a batch of %d" % (y.shape[0], NUM_TEST))
            acc_sum += (y == y_actual).astype(float).mean()
            num_sum += 1
    if num_sum == 0:
        raise ValueError("fewer than %d samples per class: nothing to score" % NUM_TEST)   # (the reference divides by zero here)
    return acc_sum / num_sum


class TemplateClassifier:
    """Stand-in for the missing frozen MNIST classifier on the "templates" synthetic digits (data_mnist.synthetic(kind="templates")):
    the nearest class pattern in the pre-sigmoid domain.  ``TemplateClassifier()`` is a predict_fn for generated_label_accuracy
    (float [n,28,28,1] in the sampler's [0, 1] range -> n class indices); as ``--label_classifier_fn
    rcgan_amd.eval_mnist:template_predict`` it serves the training CLI.  Host numpy: evaluation of a synthetic stand-in."""

    def __init__(self):
        from . import data_mnist as DM
        self.t = (DM.TEMPLATE_GAIN * DM.class_templates()).reshape(10, -1)

    def __call__(self, images):
        x = np.clip(np.asarray(images, np.float64).reshape(len(images), -1), 1e-3, 1 - 1e-3)
        a = np.log(x) - np.log1p(-x)
        d = (a * a).sum(1, keepdims=True) - 2.0 * a.dot(self.t.T) + (self.t * self.t).sum(1)[None]
        return d.argmin(1)


_TEMPLATE = None


def template_predict(images):
    """``--label_classifier_fn rcgan_amd.eval_mnist:template_predict``: one shared TemplateClassifier."""
    global _TEMPLATE
    if _TEMPLATE is None:
        _TEMPLATE = TemplateClassifier()
    return _TEMPLATE(images)
