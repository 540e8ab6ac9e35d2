#!/usr/bin/env python3
"""The reference's `examples/quickstart.rs` (BASELINE config C1) on the MI355X backend, line for line:
a 3 -> 5 -> 5 -> 1 MLP loaded from the model JSON embedded in that example (ndarray's serde wire format,
here read from examples/quickstart_model.json), a four-row labelled CSV, SGD(lr = 0.01), five epochs of
shuffled batches of two with `drop_last`, MSE (mean) loss.

    python examples/quickstart.py            # needs an MI355X; prints the loss per epoch
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CSV = """Paw_size,Tail_length,Weight,Animal
0.2,5.0,15.0,Dog
0.08,12.0,4.0,Cat
0.07,13.0,5.0,Cat
0.05,3.0,0.8,Mouse"""
LABELS = {"Dog": 1.0, "Cat": 2.0}          # anything else -> 3.0 (quickstart.rs:196-200)


def load_model(nk, dev):
    """`serde_json::from_str::<NeuralNetwork>` of quickstart.rs:53-169."""
    model = json.load(open(os.path.join(ROOT, "examples", "quickstart_model.json")))["model"]
    return [nk.serde.linear_from_json(dev, json.dumps(model[name])) for name in ("lin1", "lin2", "lin3")]


def forward(layers, x):
    out1 = layers[0].forward(x).relu()
    out2 = layers[1].forward(out1).relu()
    return layers[2].forward(out2)


def main(epochs=5, seed=0):
    import neuronika_amd
    nk = neuronika_amd.tape
    dev = nk.Device(0)
    # `from_reader_fn` maps the string label through a closure; the CSV loader here is numeric, so map it first
    rows = [line.split(",") for line in CSV.splitlines()[1:]]
    numeric = "\n".join(",".join(r[:3] + [str(LABELS.get(r[3], 3.0))]) for r in rows)
    dataset = nk.data.DataLoader().without_headers().with_labels([3]).from_string(numeric, [3], [1])
    model = load_model(nk, dev)
    optimizer = nk.optim.SGD(0.01, l2=0.0)
    for layer in model:
        optimizer.register(layer.weight)
        optimizer.register(layer.bias)
    losses = []
    for epoch in range(epochs):
        dataset.shuffle_with_seed(seed + epoch)
        total = 0.0
        for records, labels in dataset.batch(2, True):                   # .batch(2).drop_last()
            x, t = nk.from_ndarray(dev, records), nk.from_ndarray(dev, labels)
            loss = forward(model, x).mse(t, nk.Reduction.Mean)
            loss.forward()
            total += loss.item()
            loss.backward(1.0)
            optimizer.step()
            optimizer.zero_grad()
        print(f"Loss for epoch {epoch} : {total} ")
        losses.append(total)
    return losses


if __name__ == "__main__":
    main()
