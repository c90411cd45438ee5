# Control-plane image (reference Docker/dockerfile: docker-ce + conda + Azure CLI + Jupyter):
# a light image with docker CLI, Jupyter on :9999 and this repo's job tooling, for driving
# build / smoke-test / submit from a laptop or a head node.
FROM python:3.10-slim
RUN apt-get update && apt-get install -y --no-install-recommends docker.io openssh-client rsync make && \
    rm -rf /var/lib/apt/lists/* && pip install --no-cache-dir jupyter pyyaml
COPY deploy/docker/jupyter_notebook_config.py /root/.jupyter/
WORKDIR /workspace
EXPOSE 9999
CMD ["bash"]
