#!/bin/bash
# Run HERE before a gpurun call that collects profiles: the GPU box gets no .git, so the head the profiles are collected on travels as a file.
cd "$(dirname "$0")/.."
mkdir -p build
{ git rev-parse HEAD | tr -d '\n'; if ! git diff --quiet HEAD -- as_cops_and_thieves_amd bench.py; then echo -n "+uncommitted"; fi; echo; } > build/GIT_HEAD
cat build/GIT_HEAD
