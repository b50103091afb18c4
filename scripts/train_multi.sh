#!/bin/bash
# Counterpart of the reference's scripts/train_multi.sh: loop folds x cards.  Run from scripts/.
for fold in ../data/CytoSense/fold*/ ; do
  for card in ../model_cards/resnet18_cnn_2_512_clip.yaml ; do
    python3 train_multi.py -d "$fold" -m "$card"
  done
done
